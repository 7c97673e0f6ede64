// construct.hpp -- Construct_FDIFF and the small spectrum-arithmetic kernels of the FFT utilities.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_CONSTRUCT_HPP
#define SFFT_AMD_CONSTRUCT_HPP

// ------------------------------------------------------------------------------------------------
// Subtraction: kernel transfer function tables + Construct_FDIFF (SFFTConfigure.py:737-809)
//   Ctab[ij][a][m] = sum_b a_ijab W1^(m b);   Soff[ij] = sum_{ab != centre} a_ijab
//   FD[l][m] = sum_ij FI_ij[l][m] * SCALE * ( sum_a W0^(l a) Ctab[ij][a][m] - Soff[ij] )
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kernel_ctab(const double* __restrict__ sol, cplx* __restrict__ Ctab, double* __restrict__ Soff,
                                                   int Fij, int L0, int L1, int w1, int Nh, int Nhp, int N1,
                                                   const cplx* __restrict__ root1, int skip_centre)
{
    // skip_centre: separately varying scaling -- the centre coefficient a_ij00 does not multiply kernel plane ij
    // (BSplineSFFT.py:2489-2497); its term is applied in real space by scaling_term()
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int ija = blockIdx.y;                   // ij*L0 + a
    const int Fab = L0 * L1;
    if (m == 0 && (ija % L0) == 0) {              // one thread per ij
        const int ij = ija / L0;
        double sacc = 0.0;
        const int cen = (L0 / 2) * L1 + w1;
        for (int ab = 0; ab < Fab; ++ab) if (ab != cen) sacc += sol[ij * Fab + ab];
        Soff[ij] = sacc;
    }
    if (m >= Nh) return;
    const double* arow = sol + (size_t)ija * L1;  // ij*Fab + a*L1
    double cxr = 0.0, cyi = 0.0;
    const bool centre_row = skip_centre && (ija % L0) == L0 / 2;
    for (int bb = 0; bb < L1; ++bb) {
        const int b = bb - w1;
        long long q = ((long long)m * b) % N1; if (q < 0) q += N1;
        const cplx w = root1[q];
        const double av = (centre_row && b == 0) ? 0.0 : arow[bb];
        cxr = fma(av, w.x, cxr);
        cyi = fma(av, w.y, cyi);
    }
    Ctab[(size_t)ija * Nhp + m] = make_double2(cxr, cyi);
}

// separately varying scaling: DIFF -= SCALE * I * sum_s a_s00 * sbx[sp[s]][row] * sby[sq[s]][col]  (the centre term of
// Construct_FDIFF, BSplineSFFT.py:2489-2497, taken in real space: it is a plain product there)
struct ScaArgs {
    int nsca, Fab, cen;
    const double* sbx;              // [nsx][N0]
    const double* sby;              // [nsy][N1]
    int sp[SFFT_MAX_PQ], sq[SFFT_MAX_PQ];
};
__global__ void __launch_bounds__(256) scaling_term(const double* __restrict__ I, const double* __restrict__ sol, ScaArgs sa,
                                                    double* __restrict__ DIFF, int N0, int N1, double scale)
{
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= N1) return;
    double acc = 0.0;
    for (int s = 0; s < sa.nsca; ++s)
        acc = fma(sol[(size_t)s * sa.Fab + sa.cen] * sa.sbx[(size_t)sa.sp[s] * N0 + row], sa.sby[(size_t)sa.sq[s] * N1 + col], acc);
    const size_t o = (size_t)row * N1 + col;
    DIFF[o] -= scale * I[o] * acc;
}

#define CRL 8
__global__ void __launch_bounds__(256) construct_fd(const cplx* __restrict__ FI, cplx* __restrict__ FD, const cplx* __restrict__ Ctab,
                                                    const double* __restrict__ Soff, const cplx* __restrict__ root0,
                                                    int N0, int Nh, int Nhp, int Fij, int L0, int w0, double scale)
{
    __shared__ cplx wl[CRL][72];
    const int tid = threadIdx.x;
    const int m = blockIdx.x * 256 + tid;
    const int lbase = blockIdx.y * CRL;
    for (int e = tid; e < CRL * L0; e += 256) {
        const int r = e / L0, aa = e - r * L0;
        const int l = lbase + r;
        long long q = ((long long)l * (aa - w0)) % N0; if (q < 0) q += N0;
        wl[r][aa] = root0[q];
    }
    __syncthreads();
    if (m >= Nh) return;
    cplx acc[CRL];
#pragma unroll
    for (int r = 0; r < CRL; ++r) acc[r] = make_double2(0.0, 0.0);
    const size_t plane_sz = (size_t)N0 * Nhp;
    for (int ij = 0; ij < Fij; ++ij) {
        cplx kk[CRL];
#pragma unroll
        for (int r = 0; r < CRL; ++r) kk[r] = make_double2(0.0, 0.0);
        for (int aa = 0; aa < L0; ++aa) {
            const cplx c = Ctab[((size_t)ij * L0 + aa) * Nhp + m];
#pragma unroll
            for (int r = 0; r < CRL; ++r) {
                const cplx w = wl[r][aa];
                kk[r].x = fma(w.x, c.x, fma(-w.y, c.y, kk[r].x));
                kk[r].y = fma(w.x, c.y, fma(w.y, c.x, kk[r].y));
            }
        }
        const double so = Soff[ij];
#pragma unroll
        for (int r = 0; r < CRL; ++r) {
            const int l = lbase + r;
            if (l < N0) {
                const cplx fi = FI[(size_t)ij * plane_sz + (size_t)l * Nhp + m];
                const cplx kf = make_double2(scale * (kk[r].x - so), scale * kk[r].y);
                acc[r] = cadd(acc[r], cmul(fi, kf));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < CRL; ++r) {
        const int l = lbase + r;
        if (l < N0) FD[(size_t)l * Nhp + m] = acc[r];
    }
}

// ------------------------------------------------------------------------------------------------
// Small spectrum-arithmetic kernels behind the FFT utilities (noise decorrelation, FFT convolution:
// sfft/utils/PureCupyFFTKits.py, PureCupyDeCorrelationCalculator.py)
// ------------------------------------------------------------------------------------------------
__global__ void copy_spectrum_scaled(const cplx* __restrict__ src, cplx* __restrict__ dst, int N0, int Nh, int src_ld, int dst_ld, double f)
{
    const int m = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (m < Nh) { const cplx v = src[(size_t)l * src_ld + m]; dst[(size_t)l * dst_ld + m] = make_double2(v.x * f, v.y * f); }
}
__global__ void scale_real(double* __restrict__ a, double f, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] *= f;
}
// acc[i] += coeff * |a[i]|^2 * (b ? |b[i]|^2 : 1)
__global__ void spec_abs2_acc(const cplx* __restrict__ a, const cplx* __restrict__ b, double coeff, double* __restrict__ acc, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const cplx u = a[i];
    double v = coeff * (u.x * u.x + u.y * u.y);
    if (b) { const cplx w = b[i]; v *= (w.x * w.x + w.y * w.y); }
    acc[i] += v;
}
__global__ void real_rsqrt(const double* __restrict__ acc, double* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = 1.0 / sqrt(acc[i]);
}
// out[i] = a[i] * (b is complex ? b[i] : breal[i])
__global__ void spec_mul(const cplx* __restrict__ a, const cplx* __restrict__ b, const double* __restrict__ breal, cplx* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const cplx u = a[i];
    if (b) out[i] = cmul(u, b[i]);
    else { const double r = breal[i]; out[i] = make_double2(u.x * r, u.y * r); }
}
// full[l][m] of a real, conjugate-symmetric spectrum quantity from its half [N0][Nh]: full[l][N1-m] = half[(N0-l)%N0][m]
__global__ void half_to_full_real(const double* __restrict__ half, double* __restrict__ full, int N0, int N1, int Nh)
{
    const int m = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (m >= N1) return;
    full[(size_t)l * N1 + m] = (m < Nh) ? half[(size_t)l * Nh + m] : half[(size_t)((N0 - l) % N0) * Nh + (N1 - m)];
}

// debug: copy a padded half-spectrum plane to a dense [N0][Nh] array
__global__ void copy_spectrum(const cplx* __restrict__ src, cplx* __restrict__ dst, int N0, int Nh, int Nhp)
{
    const int m = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (m < Nh) dst[(size_t)l * Nh + m] = src[(size_t)l * Nhp + m];
}

#endif
