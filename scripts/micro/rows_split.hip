// Round 6: can the 4096-point ROW passes run three workgroups per CU the way the column pass now runs two?  Copies of the product's rows_r2c_4096 /
// rows_c2r_diff_4096 whose exchanges move real and imaginary parts one after the other (34.8 KB of LDS instead of 69.6 KB), compiled for three
// waves per SIMD (<= 168 registers), beside the product kernels and beside pure movers with their access pattern.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/rows_split scripts/micro/rows_split.hip && /tmp/rows_split
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

#ifndef RS_LWPS
#define RS_LWPS 3
#endif
#ifndef RS_WPS
#define RS_WPS 3
#endif
// AB: 1 no transform (mover), 2 no stores
template <int AB>
__global__ void __launch_bounds__(256, RS_WPS) rows_r2c_4096_s(RowsArgs a, RowGroups grp, cplx* __restrict__ out, int N0, int Nhp, SpecLayout lay,
                                                               const cplx* __restrict__ tw, double scale, int pairs_per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int N1 = 4096;
    const int j = threadIdx.x;
    const int pfirst = grp.first[blockIdx.y], pcount = grp.count[blockIdx.y];
    const int rp = (int)(blockIdx.x & 7) * pairs_per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= pairs_per_xcd || 2 * rp >= N0) return;
    const int l0 = 2 * rp, l1 = l0 + 1;
    const double* __restrict__ src = a.src[pfirst];
    const double* r0p = src + (size_t)l0 * N1;
    const double* r1p = src + (size_t)l1 * N1;
    double x0[16], x1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { x0[r] = ld_stream(r0p + j + 256 * r); x1[r] = ld_stream(r1p + j + 256 * r); }
    const double hs = 0.5 * scale;
    for (int pp = 0; pp < pcount; ++pp) {
        const int plane = pfirst + pp;
        const double* __restrict__ wx = a.wx[plane];
        const double* __restrict__ wy = a.wy[plane];
        const double cx0 = wx[l0], cx1 = wx[l1];
        cplx u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { const double cyp = wy[j + 256 * r]; u[r] = make_double2(x0[r] * (cx0 * cyp), x1[r] * (cx1 * cyp)); }
        if (pp > 0) __syncthreads();
        int zoff;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
        double zr[16], zi[16];          // the partners conj-pair values X[N1 - m]
        if (!(AB & 1)) {
            fft4096_core_split2(u, j, lds, tw + zoff, 0);
            __syncthreads();
#pragma unroll
            for (int sx = 0; sx < 16; ++sx) lds[j + 256 * sx] = u[R16_OUT(sx)].x;
            __syncthreads();
#pragma unroll
            for (int sx = 0; sx <= 8; ++sx) zr[sx] = lds[(N1 - (j + 256 * sx)) & (N1 - 1)];
            __syncthreads();
#pragma unroll
            for (int sx = 0; sx < 16; ++sx) lds[j + 256 * sx] = u[R16_OUT(sx)].y;
            __syncthreads();
#pragma unroll
            for (int sx = 0; sx <= 8; ++sx) zi[sx] = lds[(N1 - (j + 256 * sx)) & (N1 - 1)];
        } else {
#pragma unroll
            for (int sx = 0; sx <= 8; ++sx) { zr[sx] = u[sx].y; zi[sx] = u[sx].x; }
        }
        cplx* o0 = out + (size_t)plane * N0 * Nhp + (size_t)l0 * lay.rstride;
        cplx* o1 = out + (size_t)plane * N0 * Nhp + (size_t)l1 * lay.rstride;
#pragma unroll
        for (int sx = 0; sx <= 8; ++sx) {
            const int m = j + 256 * sx;
            if (sx < 8 || j == 0) {
                const cplx z = u[R16_OUT(sx)];
                const cplx zc = make_double2(zr[sx], -zi[sx]);
                const size_t mo = lay.col(m);
                const cplx v0 = make_double2(hs * (z.x + zc.x), hs * (z.y + zc.y)), v1 = make_double2(hs * (z.y - zc.y), -hs * (z.x - zc.x));
                if (AB & 2) { if (v0.x == 1.2345e300) st_stream(o0 + mo, v0); if (v1.x == 1.2345e300) st_stream(o1 + mo, v1); }
                else { st_stream(o0 + mo, v0); st_stream(o1 + mo, v1); }
            }
        }
    }
}


// Lean form for three workgroups per CU (<= 168 registers): the rows are NOT held across planes (planes after the first read them again, from the
// L2), the exchanges overwrite real / imaginary parts in place, and the partner exchange keeps 18 partial outputs instead of 36 partner values.
template <class T> __device__ __forceinline__ T* at_b(T* base, unsigned off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off); }
template <class T> __device__ __forceinline__ const T* at_b(const T* base, unsigned off) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off); }
template <int AB>
__global__ void __launch_bounds__(256, RS_LWPS) rows_r2c_4096_l(RowsArgs a, RowGroups grp, cplx* __restrict__ out, int N0, int Nhp, SpecLayout lay,
                                                          const cplx* __restrict__ tw, double scale, int pairs_per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int N1 = 4096;
    const int j = threadIdx.x;
    const int pfirst = grp.first[blockIdx.y], pcount = grp.count[blockIdx.y];
    const int rp = (int)(blockIdx.x & 7) * pairs_per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= pairs_per_xcd || 2 * rp >= N0) return;
    const int l0 = 2 * rp, l1 = l0 + 1;
    // addresses = (workgroup-uniform pointer) + (32-bit byte offset of the lane): see fft_r24.hpp
    const double* __restrict__ src = a.src[pfirst];
    const double* r0u = src + (size_t)l0 * N1;
    const double* r1u = src + (size_t)l1 * N1;
    const unsigned jo8 = (unsigned)j * 8u;
    const unsigned so = (unsigned)(((size_t)(j >> 2) * (size_t)lay.pstride + (size_t)(j & 3)) * sizeof(cplx));     // 4-column panels: column j + 256 sx sits 64 sx panels further
    const size_t sstep = (size_t)64 * (size_t)lay.pstride;
    const double hs = 0.5 * scale;
    for (int pp = 0; pp < pcount; ++pp) {
        const int plane = pfirst + pp;
        const double* __restrict__ wyu = a.wy[plane];
        const double cx0 = a.wx[plane][l0], cx1 = a.wx[plane][l1];
        cplx u[16];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            double f[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rr = 8 * hb + r;
                u[rr] = make_double2(*at_b(r0u + 256 * rr, jo8), *at_b(r1u + 256 * rr, jo8));
                f[r] = *at_b(wyu + 256 * rr, jo8);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { u[8 * hb + r].x *= cx0 * f[r]; u[8 * hb + r].y *= cx1 * f[r]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pp > 0) __syncthreads();
        int zoff;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
        double p0x[9], p1y[9];          // X0.x = hs (z.x + zp.x), X1.y = -hs (z.x - zp.x): need the partner's real part only
        if (!(AB & 1)) {
            fft4096_core_split2(u, j, lds, tw + zoff, 0);
            __syncthreads();
#pragma unroll
            for (int sx = 0; sx < 16; ++sx) lds[j + 256 * sx] = u[R16_OUT(sx)].x;
            __syncthreads();
            const double* pr = lds + ((N1 - j) & (N1 - 1));      // partner of element j + 256 sx: (N1 - j) - 256 sx (j = 0: element (16 - sx) 256 mod N1)
#pragma unroll
            for (int sx = 0; sx <= 8; ++sx) {
                const double zr = (j == 0) ? lds[(256 * (16 - sx)) & (N1 - 1)] : pr[-256 * sx];
                const double zx = u[R16_OUT(sx)].x; p0x[sx] = hs * (zx + zr); p1y[sx] = -hs * (zx - zr);
            }
            __syncthreads();
#pragma unroll
            for (int sx = 0; sx < 16; ++sx) lds[j + 256 * sx] = u[R16_OUT(sx)].y;
            __syncthreads();
        } else {
#pragma unroll
            for (int sx = 0; sx <= 8; ++sx) { p0x[sx] = u[sx].x; p1y[sx] = u[sx].y; }
        }
        cplx* o0u = out + (size_t)plane * N0 * Nhp + (size_t)l0 * lay.rstride;
        cplx* o1u = out + (size_t)plane * N0 * Nhp + (size_t)l1 * lay.rstride;
        const double* pr = lds + ((N1 - j) & (N1 - 1));
#pragma unroll
        for (int sx = 0; sx <= 8; ++sx) {
            if (sx < 8 || j == 0) {
                const double zy = u[R16_OUT(sx)].y;
                const double zi = (AB & 1) ? zy : ((j == 0) ? lds[(256 * (16 - sx)) & (N1 - 1)] : pr[-256 * sx]);
                const cplx v0 = make_double2(p0x[sx], hs * (zy - zi)), v1 = make_double2(hs * (zy + zi), p1y[sx]);
                cplx* q0 = at_b(o0u + sstep * sx, so); cplx* q1 = at_b(o1u + sstep * sx, so);
                if (AB & 2) { if (v0.x == 1.2345e300) st_stream(q0, v0); if (v1.x == 1.2345e300) st_stream(q1, v1); }
                else { st_stream(q0, v0); st_stream(q1, v1); }
            }
        }
    }
}

template <typename F> static float time_ms(F f, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int N0 = 4096, N1 = 4096, Nh = 2049, Nhp = 2052, reps = argc > 1 ? atoi(argv[1]) : 20;
    SpecLayout lay; lay.shift = 2; lay.mask = 3; lay.rstride = 4; lay.pstride = (long long)N0 * 4;
    const size_t P = (size_t)N0 * N1, plane_sz = (size_t)N0 * Nhp;
    std::vector<double> hI(P), hJ(P), hcy(3 * N1), hwx(N0), hones(N1, 1.0);
    srand(1);
    for (size_t i = 0; i < P; ++i) { hI[i] = rand() / (double)RAND_MAX - 0.3; hJ[i] = rand() / (double)RAND_MAX - 0.5; }
    for (int n = 0; n < N1; ++n) { const double c = (n + 1.0) / N1; hcy[n] = 1.0; hcy[N1 + n] = c; hcy[2 * N1 + n] = c * c; }
    for (int l = 0; l < N0; ++l) hwx[l] = 0.5 + (l + 1.0) / N0;
    std::vector<cplx> htw(4096);
    for (int q = 0; q < 4096; ++q) { const long double t = -2.0L * M_PIl * q / 4096.0L; htw[q] = make_double2((double)cosl(t), (double)sinl(t)); }
    double *dI, *dJ, *dcy, *dwx, *dones; cplx *dtw, *dout1, *dout2;
    HIPCHK(hipMalloc(&dI, P * 8)); HIPCHK(hipMalloc(&dJ, P * 8)); HIPCHK(hipMalloc(&dcy, 3 * N1 * 8)); HIPCHK(hipMalloc(&dwx, N0 * 8)); HIPCHK(hipMalloc(&dones, N1 * 8));
    HIPCHK(hipMalloc(&dtw, 4096 * 16)); HIPCHK(hipMalloc(&dout1, 4 * plane_sz * 16)); HIPCHK(hipMalloc(&dout2, 4 * plane_sz * 16));
    HIPCHK(hipMemcpy(dI, hI.data(), P * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dJ, hJ.data(), P * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dcy, hcy.data(), 3 * N1 * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dwx, hwx.data(), N0 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dones, hones.data(), N1 * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dtw, htw.data(), 4096 * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dout1, 0, 4 * plane_sz * 16)); HIPCHK(hipMemset(dout2, 0, 4 * plane_sz * 16));
    HIPCHK(hipFuncSetAttribute((const void*)rows_r2c_4096, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RowsArgs ra; RowGroups grp;
    for (int u = 0; u < SFFT_MAX_PLANES; ++u) { ra.src[u] = nullptr; ra.wx[u] = dones; ra.wy[u] = dones; grp.mom_out[u] = nullptr; grp.mom_nq[u] = 0; grp.first[u] = 0; grp.count[u] = 0; }
    for (int u = 0; u < 3; ++u) { ra.src[u] = dI; ra.wy[u] = dcy + (size_t)u * N1; ra.wx[u] = dwx; }
    ra.src[3] = dJ; grp.ngroups = 2; grp.first[0] = 0; grp.count[0] = 3; grp.first[1] = 3; grp.count[1] = 1;
    const int rp_per = (N0 / 2 + 7) / 8;
    const double bytes = 2.0 * P * 8 + 4 * (double)N0 * Nh * 16;
    for (int rep = 0; rep < 2; ++rep) {
        auto f_old = [&] { hipLaunchKernelGGL(rows_r2c_4096, dim3(8 * rp_per, grp.ngroups), dim3(256), F4K_LDS * sizeof(cplx), 0, ra, grp, dout1, N0, Nhp, lay, dtw, 0.25, rp_per, 0); };
        auto f_new = [&] { hipLaunchKernelGGL(rows_r2c_4096_s<0>, dim3(8 * rp_per, grp.ngroups), dim3(256), Z4K_LDS * sizeof(double), 0, ra, grp, dout2, N0, Nhp, lay, dtw, 0.25, rp_per); };
        auto f_mov = [&] { hipLaunchKernelGGL(rows_r2c_4096_s<1>, dim3(8 * rp_per, grp.ngroups), dim3(256), Z4K_LDS * sizeof(double), 0, ra, grp, dout2, N0, Nhp, lay, dtw, 0.25, rp_per); };
        auto f_nst = [&] { hipLaunchKernelGGL(rows_r2c_4096_s<2>, dim3(8 * rp_per, grp.ngroups), dim3(256), Z4K_LDS * sizeof(double), 0, ra, grp, dout2, N0, Nhp, lay, dtw, 0.25, rp_per); };
        const float t_old = time_ms(f_old, reps), t_new = time_ms(f_new, reps);
        HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
        if (rep == 0) {
            std::vector<cplx> h1(4 * plane_sz), h2(4 * plane_sz);
            HIPCHK(hipMemcpy(h1.data(), dout1, 4 * plane_sz * 16, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(h2.data(), dout2, 4 * plane_sz * 16, hipMemcpyDeviceToHost));
            double d = 0, m = 0;
            for (int pl = 0; pl < 4; ++pl) for (int l = 0; l < N0; ++l) for (int k = 0; k < Nh; ++k) {
                const size_t i = (size_t)pl * plane_sz + lay.at(l, k);
                d = std::max(d, std::max(fabs(h1[i].x - h2[i].x), fabs(h1[i].y - h2[i].y))); m = std::max(m, std::max(fabs(h1[i].x), fabs(h1[i].y)));
            }
            printf("max |product - split| / max |product| = %.3e\n", d / m);
        }
        auto f_lean = [&] { hipLaunchKernelGGL(rows_r2c_4096_l<0>, dim3(8 * rp_per, grp.ngroups), dim3(256), Z4K_LDS * sizeof(double), 0, ra, grp, dout2, N0, Nhp, lay, dtw, 0.25, rp_per); };
        auto f_lean_ns = [&] { hipLaunchKernelGGL(rows_r2c_4096_l<2>, dim3(8 * rp_per, grp.ngroups), dim3(256), Z4K_LDS * sizeof(double), 0, ra, grp, dout2, N0, Nhp, lay, dtw, 0.25, rp_per); };
        const float t_lean = time_ms(f_lean, reps);
        HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
        if (rep == 0) {
            std::vector<cplx> h1(4 * plane_sz), h2(4 * plane_sz);
            HIPCHK(hipMemcpy(h1.data(), dout1, 4 * plane_sz * 16, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(h2.data(), dout2, 4 * plane_sz * 16, hipMemcpyDeviceToHost));
            double d = 0, m = 0;
            for (int pl = 0; pl < 4; ++pl) for (int l = 0; l < N0; ++l) for (int k = 0; k < Nh; ++k) {
                const size_t i = (size_t)pl * plane_sz + lay.at(l, k);
                d = std::max(d, std::max(fabs(h1[i].x - h2[i].x), fabs(h1[i].y - h2[i].y))); m = std::max(m, std::max(fabs(h1[i].x), fabs(h1[i].y)));
            }
            printf("max |product - lean| / max |product| = %.3e\n", d / m);
        }
        const float t_lean_ns = time_ms(f_lean_ns, reps);
        printf("  lean form, three workgroups per CU: %.4f ms (%.2f TB/s), without stores %.4f\n", t_lean, bytes / t_lean * 1e-9, t_lean_ns);
        const float t_mov = time_ms(f_mov, reps), t_nst = time_ms(f_nst, reps);
        printf("rows r2c (solve launch, 4 planes): product %.4f ms (%.2f TB/s)  split exchanges, %d waves per SIMD %.4f ms (%.2f TB/s)  mover %.4f  no stores %.4f\n",
               t_old, bytes / t_old * 1e-9, RS_WPS, t_new, bytes / t_new * 1e-9, t_mov, t_nst);
    }
    return 0;
}
