// Micro-benchmark + cross-check of the round-6 row passes (fft_h2048.hpp: one real row = one 2048-point complex transform, 8 points per
// thread, four workgroups per CU) against the round-1..5 ones (fft_r16_4096.hpp: two rows per 4096-point transform, 16 points per thread,
// two workgroups per CU), in the geometry of config 2 (4096 x 4096, 4-column panels): the solve pass's row launch (3 planes of I + 1 of J,
// row moments fused), the apply pass's (3 planes of I) and the inverse row pass with its DIFF epilogue.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/rows_h2048 scripts/micro/rows_h2048.hip && /tmp/rows_h2048
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
#include "fft_h2048.hpp"

static double maxrel(const std::vector<cplx>& a, const std::vector<cplx>& b, size_t n)
{
    double d = 0, m = 0;
    for (size_t i = 0; i < n; ++i) { d = std::max(d, std::max(fabs(a[i].x - b[i].x), fabs(a[i].y - b[i].y))); m = std::max(m, std::max(fabs(a[i].x), fabs(a[i].y))); }
    return d / m;
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

int main(int argc, char** argv)
{
    const int N0 = 4096, N1 = 4096, Nh = 2049, reps = argc > 1 ? atoi(argv[1]) : 20;
    const bool rowmajor = argc > 2 && atoi(argv[2]) == 1;          // spectra as plain rows instead of 4-column panels
    const int pad = argc > 3 ? atoi(argv[3]) : 0;                   // complex elements between 4-column panels (0: panels exactly 256 KB apart)
    const int pw = argc > 4 ? atoi(argv[4]) : 4;                    // panel width in columns (a power of two)
    const int npan = (Nh + pw - 1) / pw;
    const int Nhp = npan * pw + pw * ((npan * pad + pw * N0 - 1) / (pw * N0));       // (plane stride N0 * Nhp covers the padded panels)
    SpecLayout lay; lay.shift = 0; while ((1 << lay.shift) < pw) ++lay.shift;
    lay.mask = pw - 1; lay.rstride = pw; lay.pstride = (long long)N0 * pw + pad;
    if (rowmajor) { lay.shift = 31; lay.mask = 0x7fffffff; lay.rstride = Nhp; lay.pstride = 0; }
    const size_t P = (size_t)N0 * N1, plane_sz = (size_t)N0 * Nhp;
    std::vector<double> hI(P), hJ(P), hcy(3 * N1), hwx(N0), hones(N1, 1.0);
    srand(1);
    for (size_t i = 0; i < P; ++i) { hI[i] = rand() / (double)RAND_MAX - 0.3; hJ[i] = rand() / (double)RAND_MAX - 0.5; }
    for (int n = 0; n < N1; ++n) { const double c = (n + 1.0) / N1; hcy[n] = 1.0; hcy[N1 + n] = c; hcy[2 * N1 + n] = c * c; }
    for (int l = 0; l < N0; ++l) hwx[l] = 0.5 + (l + 1.0) / N0;
    std::vector<cplx> htw(4096);
    for (int q = 0; q < 4096; ++q) { const long double t = -2.0L * M_PIl * q / 4096.0L; htw[q] = make_double2((double)cosl(t), (double)sinl(t)); }
    double *dI, *dJ, *dcy, *dwx, *dones, *dmomA, *dmomB, *dmomA2, *dmomB2, *dD1, *dD2, *dbpq, *dtbx, *dtby;
    cplx *dtw, *dout1, *dout2;
    HIPCHK(hipMalloc(&dI, P * 8)); HIPCHK(hipMalloc(&dJ, P * 8)); HIPCHK(hipMalloc(&dcy, 3 * N1 * 8)); HIPCHK(hipMalloc(&dwx, N0 * 8)); HIPCHK(hipMalloc(&dones, N1 * 8));
    HIPCHK(hipMalloc(&dmomA, (size_t)N0 * SFFT_MAX_BQ * 8)); HIPCHK(hipMalloc(&dmomB, (size_t)N0 * SFFT_MAX_BQ * 8));
    HIPCHK(hipMalloc(&dmomA2, (size_t)N0 * SFFT_MAX_BQ * 8)); HIPCHK(hipMalloc(&dmomB2, (size_t)N0 * SFFT_MAX_BQ * 8));
    HIPCHK(hipMalloc(&dD1, P * 8)); HIPCHK(hipMalloc(&dD2, P * 8));
    HIPCHK(hipMalloc(&dtw, 4096 * 16)); HIPCHK(hipMalloc(&dout1, 4 * plane_sz * 16)); HIPCHK(hipMalloc(&dout2, 4 * plane_sz * 16));
    HIPCHK(hipMemcpy(dI, hI.data(), P * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dJ, hJ.data(), P * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dcy, hcy.data(), 3 * N1 * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dwx, hwx.data(), N0 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dones, hones.data(), N1 * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dtw, htw.data(), 4096 * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dout1, 0, 4 * plane_sz * 16)); HIPCHK(hipMemset(dout2, 0, 4 * plane_sz * 16));
    // background: 6 terms (order 2), 3 column factors
    BkgArgs bk; memset(&bk, 0, sizeof(bk)); bk.npq = 6; bk.nq = 3;
    { int t = 0; for (int p_ = 0; p_ <= 2; ++p_) for (int q_ = 0; q_ <= 2 - p_; ++q_) { bk.p[t] = p_; bk.q[t] = q_; ++t; } }
    std::vector<double> hbpq = {0.3, -0.2, 0.1, 0.05, -0.07, 0.02}, htbx(3 * N0);
    for (int l = 0; l < N0; ++l) { const double c = (l + 1.0) / N0; htbx[l] = 1; htbx[N0 + l] = c; htbx[2 * N0 + l] = c * c; }
    HIPCHK(hipMalloc(&dbpq, 6 * 8)); HIPCHK(hipMalloc(&dtbx, 3 * N0 * 8)); dtby = dcy;
    HIPCHK(hipMemcpy(dbpq, hbpq.data(), 6 * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dtbx, htbx.data(), 3 * N0 * 8, hipMemcpyHostToDevice));
    bk.tbx = dtbx; bk.tby = dtby;

    HIPCHK(hipFuncSetAttribute((const void*)rows_r2c_4096, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)rows_c2r_diff_4096<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)rows_r2c_4096_h, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)rows_c2r_diff_4096_h<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    auto make = [&](bool solve, double* momI, double* momJ, RowsArgs& ra, RowGroups& grp) {
        for (int u = 0; u < SFFT_MAX_PLANES; ++u) { ra.src[u] = nullptr; ra.wx[u] = dones; ra.wy[u] = dones; grp.mom_out[u] = nullptr; grp.mom_nq[u] = 0; grp.first[u] = 0; grp.count[u] = 0; }
        for (int u = 0; u < 3; ++u) { ra.src[u] = dI; ra.wy[u] = dcy + (size_t)u * N1; ra.wx[u] = dwx; }
        grp.ngroups = 1; grp.first[0] = 0; grp.count[0] = 3;
        if (solve) { ra.src[3] = dJ; grp.ngroups = 2; grp.first[1] = 3; grp.count[1] = 1; grp.mom_out[0] = momI; grp.mom_nq[0] = 5; grp.mom_out[1] = momJ; grp.mom_nq[1] = 3; }
    };
    const int rp_per = (N0 / 2 + 7) / 8, rows_per = (N0 + 7) / 8;
    for (int solve = 1; solve >= 0; --solve) {
        RowsArgs ra1, ra2; RowGroups g1, g2;
        make(solve, dmomA, dmomB, ra1, g1); make(solve, dmomA2, dmomB2, ra2, g2);
        auto f_old = [&] { hipLaunchKernelGGL(rows_r2c_4096, dim3(8 * rp_per, g1.ngroups), dim3(256), F4K_LDS * sizeof(cplx), 0, ra1, g1, dout1, N0, Nhp, lay, dtw, 0.25, rp_per); };
        auto f_new = [&] { hipLaunchKernelGGL(rows_r2c_4096_h, dim3(8 * rows_per, g2.ngroups), dim3(256), H2K_LDS * sizeof(cplx), 0, ra2, g2, dout2, N0, Nhp, lay, dtw, 0.25, rows_per); };
        const float t_old = time_ms(f_old, reps), t_new = time_ms(f_new, reps);
        HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
        const int np = solve ? 4 : 3;
        std::vector<cplx> h1(np * plane_sz), h2(np * plane_sz);
        HIPCHK(hipMemcpy(h1.data(), dout1, np * plane_sz * 16, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(h2.data(), dout2, np * plane_sz * 16, hipMemcpyDeviceToHost));
        // compare the valid columns only (the panel's padding columns hold whatever the kernels leave there)
        double d = 0, m = 0;
        for (int pl = 0; pl < np; ++pl) for (int l = 0; l < N0; ++l) for (int k = 0; k < Nh; ++k) {
            const size_t i = (size_t)pl * plane_sz + lay.at(l, k);
            d = std::max(d, std::max(fabs(h1[i].x - h2[i].x), fabs(h1[i].y - h2[i].y))); m = std::max(m, std::max(fabs(h1[i].x), fabs(h1[i].y)));
        }
        const double bytes = (solve ? 2.0 : 1.0) * P * 8 + np * (double)N0 * Nh * 16;
        printf("rows r2c (%s, %d planes): old %.4f ms (%.2f TB/s)  new %.4f ms (%.2f TB/s)  max |old - new| / max |old| = %.3e\n", solve ? "solve" : "apply", np,
               t_old, bytes / t_old * 1e-9, t_new, bytes / t_new * 1e-9, d / m);
        if (solve) {
            std::vector<double> a1((size_t)N0 * SFFT_MAX_BQ), a2(a1.size());
            double dm = 0, mm = 0;
            for (int which = 0; which < 2; ++which) {
                HIPCHK(hipMemcpy(a1.data(), which ? dmomB : dmomA, a1.size() * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(a2.data(), which ? dmomB2 : dmomA2, a1.size() * 8, hipMemcpyDeviceToHost));
                for (int l = 0; l < N0; ++l) for (int q = 0; q < (which ? 3 : 5); ++q) { dm = std::max(dm, fabs(a1[(size_t)l * SFFT_MAX_BQ + q] - a2[(size_t)l * SFFT_MAX_BQ + q])); mm = std::max(mm, fabs(a1[(size_t)l * SFFT_MAX_BQ + q])); }
            }
            printf("  row moments: max |old - new| / max |old| = %.3e\n", dm / mm);
        }
    }
    // inverse: FD = plane 0 of the forward result (a half spectrum of a real image up to the weights: any values do)
    {
        auto f_old = [&] { hipLaunchKernelGGL(rows_c2r_diff_4096<4>, dim3((N0 + 1) / 2), dim3(256), F4K_LDS * sizeof(cplx), 0, dout1, dJ, dbpq, bk, dD1, N0, lay, dtw); };
        auto f_new = [&] { hipLaunchKernelGGL(rows_c2r_diff_4096_h<4>, dim3(8 * rows_per), dim3(256), H2K_LDS * sizeof(cplx), 0, dout1, dJ, dbpq, bk, dD2, N0, lay, dtw, rows_per); };
        const float t_old = time_ms(f_old, reps), t_new = time_ms(f_new, reps);
        HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
        std::vector<double> h1(P), h2(P);
        HIPCHK(hipMemcpy(h1.data(), dD1, P * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(h2.data(), dD2, P * 8, hipMemcpyDeviceToHost));
        double d = 0, m = 0;
        for (size_t i = 0; i < P; ++i) { d = std::max(d, fabs(h1[i] - h2[i])); m = std::max(m, fabs(h1[i])); }
        const double bytes = 2.0 * P * 8 + (double)N0 * Nh * 16;
        printf("rows c2r + DIFF: old %.4f ms (%.2f TB/s)  new %.4f ms (%.2f TB/s)  max |old - new| / max |old| = %.3e\n", t_old, bytes / t_old * 1e-9, t_new, bytes / t_new * 1e-9, d / m);
    }
    return 0;
}
