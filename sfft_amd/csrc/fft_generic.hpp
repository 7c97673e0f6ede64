// fft_generic.hpp -- generic on-chip 2-D FFT passes: rows r2c, columns c2c, rows c2r + DIFF epilogue; background evaluation helpers.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_FFT_GENERIC_HPP
#define SFFT_AMD_FFT_GENERIC_HPP

// ------------------------------------------------------------------------------------------------
// forward pass 1: rows, real -> half complex, two image rows per complex transform, SpatialPoly fused
// ------------------------------------------------------------------------------------------------
#define SFFT_MAX_PLANES 12
struct RowsArgs {                           // plane k = src[k] * wx[k][row] * wy[k][col]   (null weight = 1)
    const double* src[SFFT_MAX_PLANES];
    const double* wx[SFFT_MAX_PLANES];      // [N0] factor of the spatial basis along axis 0 (cx^i or a B-spline basis function)
    const double* wy[SFFT_MAX_PLANES];      // [N1] factor along axis 1
};

__global__ void __launch_bounds__(SFFT_FFT_MAX_THREADS) rows_r2c(RowsArgs a, cplx* __restrict__ out, int N0, int N1, int Nh, int Nhp,
                                                  SpecLayout lay, AxisDev ax, double scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int plane = blockIdx.y;
    const int l0 = 2 * blockIdx.x, l1 = l0 + 1;
    const double* __restrict__ src = a.src[plane];
    const double* __restrict__ wx = a.wx[plane];
    const double* __restrict__ wy = a.wy[plane];
    const bool has1 = l1 < N0;
    const double cx0 = wx ? wx[l0] : 1.0;
    const double cx1 = (wx && has1) ? wx[l1] : 1.0;
    {   // at most 16 elements per thread (the block has >= M / 16 threads): every load is issued before the first use
        const double* __restrict__ r0p = src + (size_t)l0 * N1;
        const double* __restrict__ r1p = src + (size_t)(has1 ? l1 : l0) * N1;
        const double h1 = has1 ? cx1 : 0.0;
        double x0[16], x1[16], wv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int n = tid + it * nt;
            const bool ok = n < N1;
            x0[it] = ok ? r0p[n] : 0.0;
            x1[it] = ok ? r1p[n] : 0.0;
            wv[it] = (ok && wy) ? wy[n] : 1.0;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int n = tid + it * nt;
            if (n < ax.M) s[n] = make_double2(x0[it] * (cx0 * wv[it]), x1[it] * (h1 * wv[it]));
        }
    }
    __syncthreads();
    lds_dft(s, ax, 1, ax.M);
    cplx* o0 = out + (size_t)plane * N0 * Nhp + (size_t)l0 * lay.rstride;
    cplx* o1 = o0 + lay.rstride;
    for (int m = tid; m < Nh; m += nt) {
        const cplx z = s[m];
        const cplx zc = cconj(s[m == 0 ? 0 : N1 - m]);
        const size_t mo = lay.col(m);
        o0[mo] = make_double2(0.5 * scale * (z.x + zc.x), 0.5 * scale * (z.y + zc.y));
        if (has1) {
            const double dx = z.x - zc.x, dy = z.y - zc.y;   // (Z - Zc) / (2i) = (dy, -dx)/2
            o1[mo] = make_double2(0.5 * scale * dy, -0.5 * scale * dx);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: columns, complex -> complex in place, TC adjacent columns per workgroup
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SFFT_FFT_MAX_THREADS) cols_c2c(cplx* __restrict__ data, int N0, int ncols, int Nhp, int TC, int LT, int MS,
                                                  SpecLayout lay, AxisDev ax, int inverse, double scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    // workgroups are dealt round-robin to the 8 XCDs: give the 8/TC tiles that share 128-byte lines to one XCD, back to back,
    // so that the partially used lines are served by that XCD's L2 (grid x is padded to a multiple of 8 * G by the launcher)
    const int G = TC >= 8 ? 1 : 8 / TC;
    const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int c0 = (((t / G) * 8 + xcd) * G + t % G) * TC;
    if (c0 >= ncols) return;
    cplx* __restrict__ base = data + (size_t)blockIdx.y * N0 * Nhp;
    // TC is a power of two and divides the block size: a thread keeps its column, its rows advance by nt / TC (no divisions and
    // one address increment per element: the loads of a thread issue back to back)
    const int c = tid & (TC - 1), lstep = nt >> LT;
    const bool cok = c0 + c < ncols;
    cplx* __restrict__ gp = base + lay.col(cok ? c0 + c : c0);
    cplx* __restrict__ sp = s + c * MS;
    const int rs = lay.rstride;
    // (a thread owns at most 16 elements -- the block has >= TC * M / 16 threads: all its loads are issued before the first use)
    {
        cplx zz[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int l = (tid >> LT) + it * lstep;
            zz[it] = (l < N0 && cok) ? gp[(size_t)(l * rs)] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int l = (tid >> LT) + it * lstep;
            if (l < ax.M) sp[l] = inverse ? cconj(zz[it]) : zz[it];
        }
    }
    __syncthreads();
    lds_dft(s, ax, TC, MS);
    if (!cok) return;
#pragma unroll 4
    for (int l = tid >> LT; l < N0; l += lstep) {
        cplx z = sp[l];
        if (inverse) z.y = -z.y;
        gp[(size_t)(l * rs)] = make_double2(z.x * scale, z.y * scale);
    }
}

// ------------------------------------------------------------------------------------------------
// forward column pass of weighted planes, out of place.  Every spatial term is  I * fx(row) * fy(col): the row pass applies fy
// only (one "stage" plane per distinct (image, column factor)), and each output plane is the column transform of its stage
// plane times fx[row].  1-D grid; on one XCD the order is: the G tiles that share 128-byte lines, then the next output of
// the same tile group -- outputs that share a stage plane find its tile in that XCD's L2.
// ------------------------------------------------------------------------------------------------
#define COLG_MAX_OUT 32
struct ColOuts {
    int nout;
    int stage_plane[COLG_MAX_OUT];                // source plane in the stage buffer
    int out_plane[COLG_MAX_OUT];                  // destination plane
    const double* wx[COLG_MAX_OUT];               // [N0] row factor
    int lo[COLG_MAX_OUT], hi[COLG_MAX_OUT];       // rows outside [lo, hi) have a zero row factor: not read (generic pass only)
};

__global__ void __launch_bounds__(SFFT_FFT_MAX_THREADS) cols_fwd_weighted(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int N0, int ncols,
                                                           int Nhp, int TC, int LT, int MS, SpecLayout lay, AxisDev ax)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int G = TC >= 8 ? 1 : 8 / TC;
    const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int gq = t % G, o = (t / G) % g.nout, tg = t / (G * g.nout);
    const int c0 = ((tg * 8 + xcd) * G + gq) * TC;
    if (c0 >= ncols) return;
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * N0 * Nhp;
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * N0 * Nhp;
    const double* __restrict__ wx = g.wx[o];
    const int c = tid & (TC - 1), lstep = nt >> LT;            // (see cols_c2c)
    const bool cok = c0 + c < ncols;
    const size_t co = lay.col(cok ? c0 + c : c0);
    const cplx* __restrict__ gp = src + co;
    cplx* __restrict__ sp = s + c * MS;
    const int rs = lay.rstride;
    {
        cplx zz[16];
        double ff[16];
        const int lo = g.lo[o], hi = g.hi[o];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int l = (tid >> LT) + it * lstep;
            const bool ok = l >= lo && l < hi && cok;
            zz[it] = ok ? gp[(size_t)(l * rs)] : make_double2(0.0, 0.0);
            ff[it] = ok ? wx[l] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int l = (tid >> LT) + it * lstep;
            if (l < ax.M) sp[l] = make_double2(zz[it].x * ff[it], zz[it].y * ff[it]);
        }
    }
    __syncthreads();
    lds_dft(s, ax, TC, MS);
    if (!cok) return;
    cplx* __restrict__ dp = dst + co;
#pragma unroll 4
    for (int l = tid >> LT; l < N0; l += lstep) dp[(size_t)(l * rs)] = sp[l];
}

// ------------------------------------------------------------------------------------------------
// inverse pass 2: rows, half complex -> real, two rows per transform, DIFF epilogue fused:
//   DIFF = J - sum_pq b_pq cx^p cy^q - conv          (SFFTSubtract.py:452-461 with the J and T terms kept in real space)
// ------------------------------------------------------------------------------------------------
#define SFFT_MAX_PQ 64
#define SFFT_MAX_BQ 16
// differential background B(row, col) = sum_t b[t] * tbx[p[t]][row] * tby[q[t]][col]  (tables of the 1-D basis factors)
struct BkgArgs {
    int npq, nq;                    // terms, distinct column factors
    const double* tbx;              // [nbx][N0]
    const double* tby;              // [nby][N1]
    int p[SFFT_MAX_PQ], q[SFFT_MAX_PQ];
};

// per-row coefficients of the column factors: c[q] = sum_{t: q[t] = q} b[t] * tbx[p[t]][row]
// (NQ = compile-time bound on the number of column factors: 4 covers polynomial backgrounds, 16 the general case)
template <int NQ>
__device__ __forceinline__ void bkg_row_coeffs(const BkgArgs& bk, const double* __restrict__ bpq, int row, int N0, double (&c)[NQ])
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) c[q] = 0.0;
    for (int t = 0; t < bk.npq; ++t) {
        const double v = bpq[t] * bk.tbx[(size_t)bk.p[t] * N0 + row];
#pragma unroll
        for (int q = 0; q < NQ; ++q) c[q] += (bk.q[t] == q) ? v : 0.0;
    }
}
template <int NQ>
__device__ __forceinline__ double bkg_eval(const BkgArgs& bk, const double (&c)[NQ], int col, int N1)
{
    double B = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double t = bk.tby[(size_t)min(q, bk.nq - 1) * N1 + col];      // clamped: always a valid, branch-free load
        B = fma((q < bk.nq) ? c[q] : 0.0, t, B);
    }
    return B;
}

template <int NQ>
__global__ void __launch_bounds__(SFFT_FFT_MAX_THREADS) rows_c2r_diff(const cplx* __restrict__ FD, const double* __restrict__ J,
                                                       const double* __restrict__ bpq, BkgArgs bk, double* __restrict__ DIFF,
                                                       int N0, int N1, int Nh, int Nhp, SpecLayout lay, AxisDev ax)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int l0 = 2 * blockIdx.x, l1 = l0 + 1;
    const bool has1 = l1 < N0;
    const cplx* f0 = FD + (size_t)l0 * lay.rstride;
    const cplx* f1 = FD + (size_t)(has1 ? l1 : l0) * lay.rstride;
    const bool even = (N1 & 1) == 0;
    {   // at most 16 elements per thread (the block has >= M / 16 threads): every load is issued before the first use
        cplx a0[16], a1[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = tid + it * nt;
            const int mm = (m >= Nh) ? N1 - m : m;
            const size_t mo = lay.col(m < N1 ? mm : 0);
            a0[it] = f0[mo];
            a1[it] = f1[mo];
        }
        const double h1 = has1 ? 1.0 : 0.0;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = tid + it * nt;
            if (m < ax.M) {
                const bool mir = m >= Nh;
                const int mm = mir ? N1 - m : m;
                cplx x0 = a0[it];
                cplx x1 = make_double2(a1[it].x * h1, a1[it].y * h1);
                if (mm == 0 || (even && mm == N1 / 2)) { x0.y = 0.0; x1.y = 0.0; }
                if (mir) { x0.y = -x0.y; x1.y = -x1.y; }
                // Z = X0 + i X1, conjugated on input so that the forward transform acts as the inverse
                s[m] = (m < N1) ? make_double2(x0.x - x1.y, -(x0.y + x1.x)) : make_double2(0.0, 0.0);
            }
        }
    }
    __syncthreads();
    lds_dft(s, ax, 1, ax.M);
    double c0[NQ], c1[NQ];
    bkg_row_coeffs<NQ>(bk, bpq, l0, N0, c0);
    bkg_row_coeffs<NQ>(bk, bpq, has1 ? l1 : l0, N0, c1);
    for (int n = tid; n < N1; n += nt) {
        const cplx z = s[n];                 // conj(result): row0 = z.x, row1 = -z.y
        DIFF[(size_t)l0 * N1 + n] = J[(size_t)l0 * N1 + n] - bkg_eval<NQ>(bk, c0, n, N1) - z.x;
        if (has1) DIFF[(size_t)l1 * N1 + n] = J[(size_t)l1 * N1 + n] - bkg_eval<NQ>(bk, c1, n, N1) + z.y;
    }
}

#endif
