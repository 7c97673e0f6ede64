// Where does the forward column pass of config 2 (cols_fwd_weighted_4096_q: 4 stage planes in, 7 spectra out, 4096-point columns, four
// columns per 512-thread workgroup, one workgroup per CU) spend its time?  The product kernel beside copies of itself with parts removed:
//   bit 1  no transforms (a pure mover with the pass's access pattern: the ceiling the memory system sets for it)
//   bit 2  no stores
//   bit 4  every workgroup reads the same tile (loads from L2: with bit 2 what remains is the on-chip time)
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/cols_q_ablate scripts/micro/cols_q_ablate.hip && /tmp/cols_q_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include <type_traits>
#include "../../include/sfft_amd.h"
typedef double2 cplx;
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#include "../../sfft_amd/csrc/device_common.hpp"
#include "../../sfft_amd/csrc/fft_generic.hpp"
#include "../../sfft_amd/csrc/fft_r16_4096.hpp"

template <int AB>
__global__ void __launch_bounds__(512) cols_q_ab(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int Nhp, SpecLayout lay,
                                                 const cplx* __restrict__ tw, int nquads)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N0 = 4096;
    const int tid = threadIdx.x, c = tid & 1, j = tid >> 1, q4 = tid & 3;
    const bool even = q4 < 2;
    const int total = nquads * g.nout;
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || logical >= total) return;
    const int cq = logical / g.nout, o = logical - cq * g.nout;
    const size_t plane_sz = (size_t)N0 * Nhp, cofs = (size_t)cq * (size_t)lay.pstride + q4;
    const size_t cofs_ld = (AB & 4) ? (size_t)(cq & 7) * (size_t)lay.pstride + q4 : cofs;
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + cofs_ld;
    const double* __restrict__ w = g.wx[o];
    const int rowA = 2 * (j >> 1);
    cplx u1[16], u2[16];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        cplx la[8], lb[8];
        double f[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            la[r] = ld_stream(src + (size_t)(rowA + 256 * (8 * hb + r)) * 4);
            lb[r] = ld_stream(src + (size_t)(rowA + 1 + 256 * (8 * hb + r)) * 4);
            f[r] = w[j + 256 * (8 * hb + r)];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const cplx got = dpp_swap2(csel(even, lb[r], la[r]));
            const cplx h1 = csel(even, la[r], got), h2 = csel(even, got, lb[r]);
            u1[8 * hb + r] = make_double2(h1.x * f[r], h1.y * f[r]);
            u2[8 * hb + r] = make_double2(h2.x * f[r], h2.y * f[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    cplx* mylds = lds + c * F4K_LDS;
    if (!(AB & 1)) {
        fft4096_core(u1, j, mylds, tw, 4 * c);
        __syncthreads();
        fft4096_core(u2, j, mylds, tw, 4 * c);
    }
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * plane_sz + cofs;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) {
        const cplx o1 = u1[R16_OUT(sx)], o2 = u2[R16_OUT(sx)];
        const cplx got = dpp_swap2(csel(even, o2, o1));
        const cplx s0 = csel(even, o1, got), s1 = csel(even, got, o2);
        if (AB & 2) {
            if (s0.x == 1.2345e300) st_stream(dst + (size_t)(rowA + 256 * sx) * 4, s0);        // (never true)
            if (s1.x == 1.2345e300) st_stream(dst + (size_t)(rowA + 1 + 256 * sx) * 4, s1);
        } else {
            st_stream(dst + (size_t)(rowA + 256 * sx) * 4, s0);
            st_stream(dst + (size_t)(rowA + 1 + 256 * sx) * 4, s1);
        }
    }
}

template <typename F> static float time_ms(F f, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int AB> static void run(const char* what, const cplx* stage, cplx* out, const ColOuts& g, int Nhp, SpecLayout lay, const cplx* tw, int nquads, int reps)
{
    HIPCHK(hipFuncSetAttribute((const void*)cols_q_ab<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int total = nquads * g.nout;
    auto f = [&] { hipLaunchKernelGGL(cols_q_ab<AB>, dim3(8 * ((total + 7) / 8)), dim3(512), (2 * F4K_LDS + 8) * sizeof(cplx), 0, stage, out, g, Nhp, lay, tw, nquads); };
    const float t = time_ms(f, reps);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    printf("%-66s %.4f ms  (11 planes = 1.477 GB: %.2f TB/s)\n", what, t, 1.477 / t);
}


// ---- candidate (round 6): ALL FOUR columns of a panel at once on 1024 threads (16 waves: four per SIMD at <= 128 registers), lane = 4 j + q4:
// a lane quad owns one 64-byte row piece outright (no DPP exchange), the four transforms run side by side, real and imaginary parts are
// exchanged one after the other through four 34 KB regions (4 x 4368 doubles = 139.8 KB).  Region q4 starts at q4 * 4368 doubles and holds
// element x at pad16(x ^ 4 q4): every 8-byte LDS access is conflict free (brute-force check: scripts/lds_conflicts.py).
#define Y4K_LDS 4368
__device__ __forceinline__ void fft4096_core_split(cplx (&u)[16], int j, double* lds, const cplx* __restrict__ tw, int sw)
{
    const int s4 = sw & 4, s8 = sw & 8;
    dft16(u);
    double* w00 = lds + 17 * j + s4 + s8;            // slot sx ^ sw = sx +- 4 +- 8 by bits 2 and 3 of sx
    double* w01 = lds + 17 * j - s4 + s8;
    double* w10 = lds + 17 * j + s4 - s8;
    double* w11 = lds + 17 * j - s4 - s8;
#define Y_W1(sx) (((sx) & 8) ? (((sx) & 4) ? w11 : w10) : (((sx) & 4) ? w01 : w00))[sx]
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) Y_W1(sx) = u[R16_OUT(sx)].x;
    __syncthreads();
    const double* rd = lds + pad16(j ^ sw);
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r].x = rd[272 * r];              // (the real parts are dead once written: overwritten in place)
    __syncthreads();
    // the imaginary parts still sit in dft16's output order: u[R16_OUT(sx)].y, untouched by the reads above
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) Y_W1(sx) = u[R16_OUT(sx)].y;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r].y = rd[272 * r];
    __syncthreads();
    const int k = j & 15;
    twiddle16(u, tw, 16 * k);
    dft16(u);
    double* w2 = lds + pad16((j - k) * 16 + (k ^ sw));
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) w2[17 * sx] = u[R16_OUT(sx)].x;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r].x = rd[272 * r];
    __syncthreads();
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) w2[17 * sx] = u[R16_OUT(sx)].y;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r].y = rd[272 * r];
    twiddle16(u, tw, j);
    dft16(u);
#undef Y_W1
}

template <int AB>
__global__ void __launch_bounds__(1024) cols_y_ab(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int Nhp, SpecLayout lay,
                                                  const cplx* __restrict__ tw, int nquads)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int N0 = 4096;
    const int tid = threadIdx.x, q4 = tid & 3, j = tid >> 2;
    const int total = nquads * g.nout;
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || logical >= total) return;
    const int cq = logical / g.nout, o = logical - cq * g.nout;
    const size_t plane_sz = (size_t)N0 * Nhp, cofs = (size_t)cq * (size_t)lay.pstride + q4;
    const size_t cofs_ld = (AB & 4) ? (size_t)(cq & 7) * (size_t)lay.pstride + q4 : cofs;
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + cofs_ld;
    const double* __restrict__ w = g.wx[o];
    cplx u[16];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {            // two batches of 8 rows: bounds the registers of the load phase
        double f[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { u[8 * hb + r] = ld_stream(src + (size_t)(j + 256 * (8 * hb + r)) * 4); f[r] = w[j + 256 * (8 * hb + r)]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) { u[8 * hb + r].x *= f[r]; u[8 * hb + r].y *= f[r]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!(AB & 1)) fft4096_core_split(u, j, lds + q4 * Y4K_LDS, tw, 4 * q4);
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * plane_sz + cofs;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) {
        const cplx v = u[R16_OUT(sx)];
        if (AB & 2) { if (v.x == 1.2345e300) st_stream(dst + (size_t)(j + 256 * sx) * 4, v); }
        else st_stream(dst + (size_t)(j + 256 * sx) * 4, v);
    }
}

template <int AB>
__global__ void __launch_bounds__(1024) cols_yp_ab(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int Nhp, SpecLayout lay,
                                                  const cplx* __restrict__ tw, int nquads)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int N0 = 4096;
    const int tid = threadIdx.x, q4 = tid & 3, j = tid >> 2;
    const int total = nquads * g.nout;
    const int per = (total + 7) >> 3;
    // persistent: workgroup b = (XCD b & 7, slot b >> 3) takes the tiles slot, slot + slots, ... of its XCD's share
    const int slots = (int)(gridDim.x >> 3);
    for (int it = (int)(blockIdx.x >> 3); it < per; it += slots) {
    const int logical = (int)(blockIdx.x & 7) * per + it;
    if (logical >= total) break;
    if (it != (int)(blockIdx.x >> 3)) __syncthreads();          // the previous tile's last LDS reads are done
    __builtin_amdgcn_sched_barrier(0);
    const int cq = logical / g.nout, o = logical - cq * g.nout;
    const size_t plane_sz = (size_t)N0 * Nhp, cofs = (size_t)cq * (size_t)lay.pstride + q4;
    const size_t cofs_ld = (AB & 4) ? (size_t)(cq & 7) * (size_t)lay.pstride + q4 : cofs;
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + cofs_ld;
    const double* __restrict__ w = g.wx[o];
    cplx u[16];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {            // two batches of 8 rows: bounds the registers of the load phase
        double f[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { u[8 * hb + r] = ld_stream(src + (size_t)(j + 256 * (8 * hb + r)) * 4); f[r] = w[j + 256 * (8 * hb + r)]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) { u[8 * hb + r].x *= f[r]; u[8 * hb + r].y *= f[r]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    int zoff;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));          // (keeps the stage twiddles inside the tile loop)
    if (!(AB & 1)) fft4096_core_split(u, j, lds + q4 * Y4K_LDS, tw + zoff, 4 * q4);
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * plane_sz + cofs;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) {
        const cplx v = u[R16_OUT(sx)];
        if (AB & 2) { if (v.x == 1.2345e300) st_stream(dst + (size_t)(j + 256 * sx) * 4, v); }
        else st_stream(dst + (size_t)(j + 256 * sx) * 4, v);
    }
}
}

template <int AB> static void run_y(const char* what, const cplx* stage, cplx* out, const ColOuts& g, int Nhp, SpecLayout lay, const cplx* tw, int nquads, int reps)
{
    HIPCHK(hipFuncSetAttribute((const void*)cols_y_ab<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int total = nquads * g.nout;
    auto f = [&] { hipLaunchKernelGGL(cols_y_ab<AB>, dim3(8 * ((total + 7) / 8)), dim3(1024), 4 * Y4K_LDS * sizeof(double), 0, stage, out, g, Nhp, lay, tw, nquads); };
    const float t = time_ms(f, reps);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    printf("Y  %-63s %.4f ms  (11 planes = 1.477 GB: %.2f TB/s)\n", what, t, 1.477 / t);
}

template <int AB> static void run_yp(const char* what, const cplx* stage, cplx* out, const ColOuts& g, int Nhp, SpecLayout lay, const cplx* tw, int nquads, int reps, int wgs)
{
    HIPCHK(hipFuncSetAttribute((const void*)cols_yp_ab<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto f = [&] { hipLaunchKernelGGL(cols_yp_ab<AB>, dim3(wgs), dim3(1024), 4 * Y4K_LDS * sizeof(double), 0, stage, out, g, Nhp, lay, tw, nquads); };
    const float t = time_ms(f, reps);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    printf("YP %-63s %.4f ms  (11 planes = 1.477 GB: %.2f TB/s)\n", what, t, 1.477 / t);
}


// ---- candidate Z (round 6): TWO workgroups per CU.  A 512-thread workgroup (lane = 2 j + c, <= 128 registers, 2 x 34 KB of LDS) transforms ONE
// column pair; what makes that affordable is the layout on both sides: the stage planes keep their 4-column panels but each 128-byte line
// (rows 2p, 2p + 1 x columns 0..3) is stored pair-major -- [pair h][row parity][column c] -- so a lane quad reads one whole 64-byte sector
// (the row pass, which writes whole lines, does not care); the spectra go out in 2-column panels [Nhp/2][N0][2]: a wave stores 1 KB contiguous.
__device__ __forceinline__ void fft4096_core_split2(cplx (&u)[16], int j, double* lds, const cplx* __restrict__ tw, int sw)     // sw = 0 or 8
{
    dft16(u);
    double* wA = lds + 17 * j + sw;                  // slot sx ^ sw
    double* wB = lds + 17 * j - sw;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) ((sx & 8) ? wB : wA)[sx] = u[R16_OUT(sx)].x;
    __syncthreads();
    const double* rd = lds + pad16(j ^ sw);
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r].x = rd[272 * r];
    __syncthreads();
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) ((sx & 8) ? wB : wA)[sx] = u[R16_OUT(sx)].y;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r].y = rd[272 * r];
    __syncthreads();
    const int k = j & 15;
    twiddle16(u, tw, 16 * k);
    dft16(u);
    double* w2 = lds + pad16((j - k) * 16 + (k ^ sw));
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) w2[17 * sx] = u[R16_OUT(sx)].x;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r].x = rd[272 * r];
    __syncthreads();
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) w2[17 * sx] = u[R16_OUT(sx)].y;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r].y = rd[272 * r];
    twiddle16(u, tw, j);
    dft16(u);
}

// standard 4-column panels -> pair-major lines (test helper: the product's row pass would write this directly)
__global__ void permute_p4x(const cplx* __restrict__ in, cplx* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t line = i >> 3; const int w = (int)(i & 7), par = w >> 2, h = (w >> 1) & 1, c = w & 1;        // standard: [row parity][h][c]
    out[line * 8 + h * 4 + par * 2 + c] = in[i];
}

template <int AB>
__global__ void __launch_bounds__(512, 4) cols_z_ab(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int Nhp, long long pstride4,
                                                    const cplx* __restrict__ tw, int npairs)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int N0 = 4096;
    const int tid = threadIdx.x, c = tid & 1, j = tid >> 1;
    const int total = npairs * g.nout;
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || logical >= total) return;
    // (output fastest, then the two pairs of a panel: the 14 workgroups that touch one stage panel run back to back on one XCD)
    const int cp = logical / g.nout, o = logical - cp * g.nout;
    const int cpl = (AB & 4) ? (cp & 15) : cp;
    const size_t plane_sz = (size_t)N0 * Nhp;
    // element (row l, pair h of panel P, column c) of a stage plane: P * pstride4 + (l >> 1) * 8 + h * 4 + (l & 1) * 2 + c
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + (size_t)(cpl >> 1) * (size_t)pstride4 + (size_t)((cpl & 1) * 4)
                                   + (size_t)((tid >> 2) * 8 + (tid & 3));
    const double* __restrict__ w = g.wx[o];
    cplx u[16];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        double f[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { u[8 * hb + r] = ld_stream(src + (size_t)(1024 * (8 * hb + r))); f[r] = w[j + 256 * (8 * hb + r)]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) { u[8 * hb + r].x *= f[r]; u[8 * hb + r].y *= f[r]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!(AB & 1)) fft4096_core_split2(u, j, lds + c * Y4K_LDS, tw, 8 * c);
    // 2-column panels: element (row l, column c of pair cp) at cp * N0 * 2 + l * 2 + c
    // (AB & 8: the spectra in pair-major 4-column panels as well -- the sibling pair's workgroup writes the other half of every line)
    cplx* __restrict__ dst = (AB & 8) ? out + (size_t)g.out_plane[o] * plane_sz + (size_t)(cp >> 1) * (size_t)pstride4 + (size_t)((cp & 1) * 4) + (size_t)((tid >> 2) * 8 + (tid & 3))
                                      : out + (size_t)g.out_plane[o] * plane_sz + (size_t)cp * (size_t)(N0 * 2) + tid;
    const int sstep = (AB & 8) ? 1024 : 512;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) {
        const cplx v = u[R16_OUT(sx)];
        if (AB & 2) { if (v.x == 1.2345e300) st_stream(dst + sstep * sx, v); }
        else st_stream(dst + sstep * sx, v);
    }
}

template <int AB> static void run_z(const char* what, const cplx* stage, cplx* out, const ColOuts& g, int Nhp, long long pstride4, const cplx* tw, int npairs, int reps)
{
    HIPCHK(hipFuncSetAttribute((const void*)cols_z_ab<AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int total = npairs * g.nout;
    auto f = [&] { hipLaunchKernelGGL(cols_z_ab<AB>, dim3(8 * ((total + 7) / 8)), dim3(512), 2 * Y4K_LDS * sizeof(double), 0, stage, out, g, Nhp, pstride4, tw, npairs); };
    const float t = time_ms(f, reps);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    printf("Z  %-63s %.4f ms  (11 planes = 1.477 GB: %.2f TB/s)\n", what, t, 1.477 / t);
}

int main(int argc, char** argv)
{
    const int N0 = 4096, Nh = 2049, Nhp = 2052, reps = argc > 1 ? atoi(argv[1]) : 20;
    SpecLayout lay; lay.shift = 2; lay.mask = 3; lay.rstride = 4; lay.pstride = (long long)N0 * 4;
    const size_t plane_sz = (size_t)N0 * Nhp;
    cplx *dstage, *dout, *dtw; double* dwx;
    HIPCHK(hipMalloc(&dstage, 4 * plane_sz * 16)); HIPCHK(hipMalloc(&dout, 7 * plane_sz * 16)); HIPCHK(hipMalloc(&dtw, 4096 * 16)); HIPCHK(hipMalloc(&dwx, 3 * N0 * 8));
    std::vector<cplx> hs(plane_sz), htw(4096);
    srand(2);
    for (size_t i = 0; i < plane_sz; ++i) hs[i] = make_double2(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
    for (int k = 0; k < 4; ++k) HIPCHK(hipMemcpy(dstage + k * plane_sz, hs.data(), plane_sz * 16, hipMemcpyHostToDevice));
    for (int q = 0; q < 4096; ++q) { const long double t = -2.0L * M_PIl * q / 4096.0L; htw[q] = make_double2((double)cosl(t), (double)sinl(t)); }
    HIPCHK(hipMemcpy(dtw, htw.data(), 4096 * 16, hipMemcpyHostToDevice));
    std::vector<double> hwx(3 * N0);
    for (int l = 0; l < N0; ++l) { const double c = (l + 1.0) / N0; hwx[l] = 1; hwx[N0 + l] = c; hwx[2 * N0 + l] = c * c; }
    HIPCHK(hipMemcpy(dwx, hwx.data(), 3 * N0 * 8, hipMemcpyHostToDevice));
    ColOuts g; memset(&g, 0, sizeof(g));
    const int sp[7] = {0, 0, 0, 1, 1, 2, 3}, wi[7] = {0, 1, 2, 0, 1, 0, 0};       // stage-major, as the launcher orders them
    g.nout = 7;
    for (int o = 0; o < 7; ++o) { g.stage_plane[o] = sp[o]; g.out_plane[o] = o; g.wx[o] = dwx + (size_t)wi[o] * N0; }
    const int nquads = (Nh + 3) / 4;
    HIPCHK(hipFuncSetAttribute((const void*)cols_fwd_weighted_4096_q, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    {
        const int total = nquads * g.nout;
        auto f = [&] { hipLaunchKernelGGL(cols_fwd_weighted_4096_q, dim3(8 * ((total + 7) / 8)), dim3(512), (2 * F4K_LDS + 8) * sizeof(cplx), 0, dstage, dout, g, Nhp, lay, dtw, nquads); };
        const float t = time_ms(f, reps);
        printf("%-66s %.4f ms  (11 planes = 1.477 GB: %.2f TB/s)\n", "product kernel", t, 1.477 / t);
    }
    run<0>("copy of it", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    run<1>("no transforms (mover with this access pattern)", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    run<2>("no stores", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    run<3>("no transforms, no stores (loads only)", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    run<4>("loads from 8 tiles (L2)", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    run<6>("loads from 8 tiles, no stores (on-chip time)", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    run<5>("loads from 8 tiles, no transforms (stores only)", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    // candidate: cross-check against the product kernel first
    {
        const int total = nquads * g.nout;
        cplx* dout2; HIPCHK(hipMalloc(&dout2, 7 * plane_sz * 16));
        hipLaunchKernelGGL(cols_fwd_weighted_4096_q, dim3(8 * ((total + 7) / 8)), dim3(512), (2 * F4K_LDS + 8) * sizeof(cplx), 0, dstage, dout, g, Nhp, lay, dtw, nquads);
        HIPCHK(hipFuncSetAttribute((const void*)cols_y_ab<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(cols_y_ab<0>, dim3(8 * ((total + 7) / 8)), dim3(1024), 4 * Y4K_LDS * sizeof(double), 0, dstage, dout2, g, Nhp, lay, dtw, nquads);
        HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
        std::vector<cplx> a(plane_sz), b(plane_sz);
        double d = 0, m = 0;
        for (int pl = 0; pl < 7; ++pl) {
            HIPCHK(hipMemcpy(a.data(), dout + pl * plane_sz, plane_sz * 16, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(b.data(), dout2 + pl * plane_sz, plane_sz * 16, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < plane_sz; ++i) { d = std::max(d, std::max(fabs(a[i].x - b[i].x), fabs(a[i].y - b[i].y))); m = std::max(m, std::max(fabs(a[i].x), fabs(a[i].y))); }
        }
        printf("Y  max |product - candidate| / max |product| over the 7 planes = %.3e\n", d / m);
        HIPCHK(hipFree(dout2));
    }
    run_y<0>("candidate: four columns side by side on 1024 threads", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    run_y<1>("no transforms (mover)", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    run_y<2>("no stores", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    run_y<6>("loads from 8 tiles, no stores (on-chip time)", dstage, dout, g, Nhp, lay, dtw, nquads, reps);
    {
        const int total = nquads * g.nout;
        cplx* dout2; HIPCHK(hipMalloc(&dout2, 7 * plane_sz * 16));
        hipLaunchKernelGGL(cols_fwd_weighted_4096_q, dim3(8 * ((total + 7) / 8)), dim3(512), (2 * F4K_LDS + 8) * sizeof(cplx), 0, dstage, dout, g, Nhp, lay, dtw, nquads);
        HIPCHK(hipFuncSetAttribute((const void*)cols_yp_ab<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(cols_yp_ab<0>, dim3(256), dim3(1024), 4 * Y4K_LDS * sizeof(double), 0, dstage, dout2, g, Nhp, lay, dtw, nquads);
        HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
        std::vector<cplx> a(plane_sz), b(plane_sz);
        double d = 0, m = 0;
        for (int pl = 0; pl < 7; ++pl) {
            HIPCHK(hipMemcpy(a.data(), dout + pl * plane_sz, plane_sz * 16, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(b.data(), dout2 + pl * plane_sz, plane_sz * 16, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < plane_sz; ++i) { d = std::max(d, std::max(fabs(a[i].x - b[i].x), fabs(a[i].y - b[i].y))); m = std::max(m, std::max(fabs(a[i].x), fabs(a[i].y))); }
        }
        printf("YP max |product - persistent candidate| / max |product| = %.3e\n", d / m);
        HIPCHK(hipFree(dout2));
    }
    {
        // candidate Z: pair-major stage lines in, 2-column panels out
        const int total4 = nquads * g.nout, npairs = (Nh + 1) / 2;
        cplx *dstx, *dout2; HIPCHK(hipMalloc(&dstx, 4 * plane_sz * 16)); HIPCHK(hipMalloc(&dout2, 7 * plane_sz * 16));
        HIPCHK(hipMemset(dout2, 0, 7 * plane_sz * 16));
        hipLaunchKernelGGL(permute_p4x, dim3((unsigned)((4 * plane_sz + 255) / 256)), dim3(256), 0, 0, dstage, dstx, 4 * plane_sz);
        hipLaunchKernelGGL(cols_fwd_weighted_4096_q, dim3(8 * ((total4 + 7) / 8)), dim3(512), (2 * F4K_LDS + 8) * sizeof(cplx), 0, dstage, dout, g, Nhp, lay, dtw, nquads);
        HIPCHK(hipFuncSetAttribute((const void*)cols_z_ab<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const int totalz = npairs * g.nout;
        hipLaunchKernelGGL(cols_z_ab<0>, dim3(8 * ((totalz + 7) / 8)), dim3(512), 2 * Y4K_LDS * sizeof(double), 0, dstx, dout2, g, Nhp, lay.pstride, dtw, npairs);
        HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
        std::vector<cplx> a(plane_sz), b(plane_sz);
        double d = 0, m = 0;
        for (int pl = 0; pl < 7; ++pl) {
            HIPCHK(hipMemcpy(a.data(), dout + pl * plane_sz, plane_sz * 16, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(b.data(), dout2 + pl * plane_sz, plane_sz * 16, hipMemcpyDeviceToHost));
            for (int l = 0; l < N0; ++l) for (int k = 0; k < Nh; ++k) {
                const cplx x = a[lay.at(l, k)], y = b[(size_t)(k >> 1) * N0 * 2 + (size_t)l * 2 + (k & 1)];
                d = std::max(d, std::max(fabs(x.x - y.x), fabs(x.y - y.y))); m = std::max(m, std::max(fabs(x.x), fabs(x.y)));
            }
        }
        printf("Z  max |product - candidate Z| / max |product| over the 7 planes = %.3e\n", d / m);
        run_z<0>("candidate Z: one column pair per 512-thread workgroup, 2 per CU", dstx, dout2, g, Nhp, lay.pstride, dtw, npairs, reps);
        run_z<1>("no transforms (mover)", dstx, dout2, g, Nhp, lay.pstride, dtw, npairs, reps);
        run_z<2>("no stores", dstx, dout2, g, Nhp, lay.pstride, dtw, npairs, reps);
        run_z<6>("loads from 16 tiles, no stores (on-chip time)", dstx, dout2, g, Nhp, lay.pstride, dtw, npairs, reps);
        {   // pair-major lines out: check, then time
            HIPCHK(hipFuncSetAttribute((const void*)cols_z_ab<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipMemset(dout2, 0, 7 * plane_sz * 16));
            hipLaunchKernelGGL(cols_z_ab<8>, dim3(8 * ((totalz + 7) / 8)), dim3(512), 2 * Y4K_LDS * sizeof(double), 0, dstx, dout2, g, Nhp, lay.pstride, dtw, npairs);
            HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
            double d2 = 0, m2 = 0;
            for (int pl = 0; pl < 7; ++pl) {
                HIPCHK(hipMemcpy(a.data(), dout + pl * plane_sz, plane_sz * 16, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(b.data(), dout2 + pl * plane_sz, plane_sz * 16, hipMemcpyDeviceToHost));
                for (int l = 0; l < N0; ++l) for (int k = 0; k < Nh; ++k) {
                    const cplx x = a[lay.at(l, k)], y = b[(size_t)(k >> 2) * lay.pstride + (size_t)(l >> 1) * 8 + ((k >> 1) & 1) * 4 + (l & 1) * 2 + (k & 1)];
                    d2 = std::max(d2, std::max(fabs(x.x - y.x), fabs(x.y - y.y))); m2 = std::max(m2, std::max(fabs(x.x), fabs(x.y)));
                }
            }
            printf("Z8 max |product - candidate Z, pair-major lines out| / max |product| = %.3e\n", d2 / m2);
        }
        run_z<8>("candidate Z, pair-major 4-column panels out", dstx, dout2, g, Nhp, lay.pstride, dtw, npairs, reps);
        run_z<9>("  no transforms (mover)", dstx, dout2, g, Nhp, lay.pstride, dtw, npairs, reps);
        run_z<0>("candidate Z again (2-column panels out)", dstx, dout2, g, Nhp, lay.pstride, dtw, npairs, reps);
        run_z<8>("candidate Z, pair-major out, again", dstx, dout2, g, Nhp, lay.pstride, dtw, npairs, reps);
        HIPCHK(hipFree(dstx)); HIPCHK(hipFree(dout2));
    }
    return 0;
}
