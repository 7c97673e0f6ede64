// fft_r24_6144.hpp -- register-resident fast path for 6144-point column axes (6144 = 16 * 16 * 24).
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_FFT_R24_6144_HPP
#define SFFT_AMD_FFT_R24_6144_HPP

// ================================================================================================
// The generic column pass keeps a whole 6144-point column in LDS (98 KB): one workgroup per CU, whose load, transform and store
// phases never overlap, and ten trips of the column through LDS.  Here the column lives in registers as in fft_r16_4096.hpp:
// 384 threads own 16 points each through two radix-16 stages, 256 of them own 24 points each in a last radix-24 stage, and LDS
// only carries the two exchanges between stages -- real and imaginary parts one after the other, so that a workgroup needs
// 52 KB and two of them share a CU (one loads or stores while the other computes).
//
// Index algebra (n = j + 384 r, k = s + 16 (t + 16 k3); W = exp(-2 pi i / 6144)):
//   stage 1, thread j < 384:            A[j][s]      = sum_{r < 16} x[j + 384 r] W16^(r s)
//   stage 2, thread 16 jl + s, jl < 24: B[jl][s][t]  = sum_{r < 16} W^(24 s r) A[jl + 24 r][s] W16^(r t)
//   stage 3, thread q = s + 16 t < 256: X[q + 256 k3] = sum_{jl < 24} W^(jl q) B[jl][s][t] W24^(jl k3)
// ================================================================================================
#define F6K_LDS 6528                                 // doubles: 6144 + 6144 / 16 (the padded layout of pad16)

// forward 24-point DFT in registers (24 = 3 * 8: n = 3 b + a, k = d + 8 c), natural order in and out
__device__ __forceinline__ void dft24(const cplx (&x)[24], cplx (&X)[24])
{
    // W24^m = (C[m], -S[m]), m = a d <= 14
    constexpr double C[15] = { 1.0, 0.96592582628906828675, 0.86602540378443864676, 0.70710678118654752440, 0.5, 0.25881904510252076235, 0.0,
                               -0.25881904510252076235, -0.5, -0.70710678118654752440, -0.86602540378443864676, -0.96592582628906828675, -1.0,
                               -0.96592582628906828675, -0.86602540378443864676 };
    constexpr double S[15] = { 0.0, 0.25881904510252076235, 0.5, 0.70710678118654752440, 0.86602540378443864676, 0.96592582628906828675, 1.0,
                               0.96592582628906828675, 0.86602540378443864676, 0.70710678118654752440, 0.5, 0.25881904510252076235, 0.0,
                               -0.25881904510252076235, -0.5 };
    const double S3 = 0.86602540378443864676;
    cplx G[3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 8; ++b) G[a][b] = x[3 * b + a];
        dft8(G[a]);
        if (a > 0) {
#pragma unroll
            for (int d = 1; d < 8; ++d) G[a][d] = cmul(G[a][d], make_double2(C[a * d], -S[a * d]));
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const cplx u0 = G[0][d], u1 = G[1][d], u2 = G[2][d];
        const cplx sm = cadd(u1, u2), df = csub(u1, u2);
        const cplx m = make_double2(u0.x - 0.5 * sm.x, u0.y - 0.5 * sm.y);
        X[d] = cadd(u0, sm);
        X[d + 8] = make_double2(m.x + S3 * df.y, m.y - S3 * df.x);
        X[d + 16] = make_double2(m.x - S3 * df.y, m.y + S3 * df.x);
    }
}

// 6144-point forward FFT.  In: u[r] = x[j + 384 r], j < 384.  Out, threads j < 256 only: v[k3] = X[j + 256 k3], k3 < 24.
// `lds` = F6K_LDS doubles.  Every thread of the 384-thread block must call (barriers inside); tw[k] = W^k, k < 6144.
__device__ __forceinline__ void fft6144_core(cplx (&u)[16], cplx (&v)[24], int j, double* lds, const cplx* __restrict__ tw)
{
    const int jp = j + (j >> 4);                     // pad16(j + 384 r) = jp + 408 r,  pad16(j + 256 r) = jp + 272 r
    dft16(u);
    // exchange 1: A[j][s] sits at element 16 j + s; the stage-2 thread j = 16 jl + s takes elements j + 384 r
    {
        double re[16];
#pragma unroll
        for (int sx = 0; sx < 16; ++sx) lds[17 * j + sx] = u[R16_OUT(sx)].x;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) re[r] = lds[jp + 408 * r];
        __syncthreads();
#pragma unroll
        for (int sx = 0; sx < 16; ++sx) lds[17 * j + sx] = u[R16_OUT(sx)].y;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = make_double2(re[r], lds[jp + 408 * r]);
        __syncthreads();
    }
    const int s = j & 15, jl = j >> 4;
    twiddle16(u, tw, 24 * s);
    dft16(u);
    // exchange 2: B[jl][s][t] goes to element 256 jl + s + 16 t (padded: 272 jl + 17 t + s); the stage-3 thread q takes q + 256 r
    {
        double re[24];
        const int wb = 272 * jl + s;
#pragma unroll
        for (int tx = 0; tx < 16; ++tx) lds[wb + 17 * tx] = u[R16_OUT(tx)].x;
        __syncthreads();
        if (j < 256) {
#pragma unroll
            for (int r = 0; r < 24; ++r) re[r] = lds[jp + 272 * r];
        }
        __syncthreads();
#pragma unroll
        for (int tx = 0; tx < 16; ++tx) lds[wb + 17 * tx] = u[R16_OUT(tx)].y;
        __syncthreads();
        if (j >= 256) return;
        cplx xin[24];
#pragma unroll
        for (int r = 0; r < 24; ++r) xin[r] = make_double2(re[r], lds[jp + 272 * r]);
#pragma unroll
        for (int r = 1; r < 24; ++r) xin[r] = cmul(xin[r], tw[r * j]);
        dft24(xin, v);
    }
}

// Forward column pass of the weighted planes for N0 = 6144 (see cols_fwd_weighted: same arguments, same XCD-aware order -- on one
// XCD the eight columns that share 128-byte lines, then the next output of the same column group).  One column per workgroup.
__global__ void __launch_bounds__(384) cols_fwd_weighted_6144(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int ncols,
                                                              int Nhp, SpecLayout lay, const cplx* __restrict__ tw)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int N0 = 6144;
    const int j = threadIdx.x;
    const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int gq = t & 7, o = (t >> 3) % g.nout, tg = (t >> 3) / g.nout;
    const int col = (tg * 8 + xcd) * 8 + gq;
    if (col >= ncols) return;
    const size_t plane_sz = (size_t)N0 * Nhp, cofs = lay.col(col), rs = (size_t)lay.rstride;
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + cofs;
    const double* __restrict__ w = g.wx[o];
    const int lo = g.lo[o], hi = g.hi[o];
    cplx u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int l = j + 384 * r;
        const bool ok = l >= lo && l < hi;
        const cplx z = ok ? src[(size_t)l * rs] : make_double2(0.0, 0.0);
        const double f = ok ? w[l] : 0.0;
        u[r] = make_double2(z.x * f, z.y * f);
    }
    cplx v[24];
    fft6144_core(u, v, j, lds, tw);
    if (j >= 256) return;
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * plane_sz + cofs;
#pragma unroll
    for (int k3 = 0; k3 < 24; ++k3) dst[(size_t)(j + 256 * k3) * rs] = v[k3];
}

#endif
