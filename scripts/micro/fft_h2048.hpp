// fft_h2048.hpp -- 4096-point REAL rows as ONE 2048-point complex transform per row, 8 points per thread.
// EXPERIMENT of round 6 (scripts/micro/rows_h2048.hip is its only user; results in docs/LOG.md): correct to 7e-17 against the product's
// row passes, NOT faster (0.255 against 0.240 ms forward, 0.112 against 0.113 ms inverse), so the product keeps fft_r16_4096.hpp.
#ifndef SFFT_AMD_FFT_H2048_HPP
#define SFFT_AMD_FFT_H2048_HPP

// ================================================================================================
// Why (round 6): the radix-16 row passes of fft_r16_4096.hpp pack TWO real rows into one 4096-point transform held by 256
// threads x 16 points: 204 - 240 registers (two waves per SIMD), 69.6 KB of LDS (two workgroups per CU), and with all HBM traffic
// removed they still take 76 - 79 % of their time (profiles/r05_o_cache_resident_bound.txt): every CU holds two workgroups whose
// load -> 16-point -> exchange -> 16-point -> exchange -> 16-point -> partner exchange -> store phases hardly overlap.
// Here a real row x[0..4096) is the complex sequence z[m] = x[2m] + i x[2m+1] (m < 2048): ONE row per 256-thread workgroup,
// 8 points per thread (<= 128 registers: four waves per SIMD), 36 KB of LDS (four workgroups per CU, four rows in flight per CU
// as before but in four independent phases and on twice the waves), 16-byte loads of the image, and no partner exchange:
// the last stage leaves Z[k] and Z[2048 - k] in the SAME thread.
//
// Network (decimation in frequency, 2048 = 8 x 8 x 8 x 4; thread j in [0, 256); tw[q] = exp(-2 pi i q / 4096)):
//   stage 1   u[r] = z[j + 256 r]           dft8 -> s1, times tw[2 j s1]
//   E1        element (j, s1) at 256 s1 + j;            reader (a = j & 31, s1 = j >> 5) takes (a + 32 b, s1), b < 8
//   stage 2   dft8 over b -> s2, times tw[16 a s2]
//   E2        element (a, Q = s1 + 8 s2) at a + 36 Q;   reader (c = j & 3, Q = j >> 2) takes (c + 4 d, Q), d < 8
//   stage 3   dft8 over d -> s3, times tw[128 c s3]
//   E3        element (c, C = Q + 64 s3) at P3 c + C;   reader j takes the combos C = j and C' = 512 - j (256 for j = 0), c < 4
//   stage 4   two dft4 over c -> s4:  u[s4] = Z[j + 512 s4],  u[4 + s4] = Z[C' + 512 s4]
// 2048 - (j + 512 s4) = C' + 512 (3 - s4): the partner of u[s4] is u[7 - s4] (thread 0: combos 0 and 256 are their own partners).
// The transposed network (fft2048_tr: the same stages in reverse order, twiddles before the butterflies) takes that ownership as
// its INPUT and leaves FFT(G)[j + 256 r] in u[r]: the DFT matrix is symmetric, so it is the same transform -- the inverse row pass.
// Every 16-byte LDS access of both directions is conflict free (scripts/lds_conflicts.py; E3 needs P3 = 514 when written by
// (c, Q) -- the forward network -- and 516 when read by (c, Q)).  Index algebra: tests/test_hip_model.py (numpy model).
// ================================================================================================
#define H2K_LDS 2304                                 // complex elements of LDS per transform (E2: 31 + 36 * 63 = 2299 is the largest index)
#define H2K_P3F 514
#define H2K_P3T 516
#ifndef H2K_RELOAD
#define H2K_RELOAD 0
#endif
#ifndef H2K_ABLATE
#define H2K_ABLATE 0                                 // micro-benchmark ablations (scripts/micro/rows_h2048.hip): 1 no twiddle loads, 2 no column-factor loads, 4 no transform, 8 no stores, 16 one image load per thread
#endif
#ifndef H2K_WPS
#define H2K_WPS 4                                    // waves per SIMD the forward kernel is compiled for (4: <= 128 registers)
#endif

// u[s] *= tw[s q], s = 1..7, from three table entries (4 q < 4096)
__device__ __forceinline__ void twiddle8(cplx (&u)[8], const cplx* __restrict__ tw, int q)
{
#if H2K_ABLATE & 1
    const cplx w1 = make_double2(0.6, -0.8 + 1e-9 * q), w2 = make_double2(0.8, -0.6), w4 = make_double2(1.0, 1e-9 * q);
#else
    const cplx w1 = tw[q], w2 = tw[2 * q], w4 = tw[4 * q];
#endif
    const cplx w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2);
    const cplx w7 = cmul(w4, w3);
    u[1] = cmul(u[1], w1); u[2] = cmul(u[2], w2); u[3] = cmul(u[3], w3); u[4] = cmul(u[4], w4);
    u[5] = cmul(u[5], w5); u[6] = cmul(u[6], w6); u[7] = cmul(u[7], w7);
}

// In: u[r] = z[j + 256 r].  Out: u[s4] = Z[j + 512 s4], u[4 + s4] = Z[C' + 512 s4], C' = j ? 512 - j : 256.  Every thread of the
// 256-thread block must call (barriers inside); the caller puts a barrier between the last LDS read here and its next LDS write.
__device__ __forceinline__ void fft2048_fwd(cplx (&u)[8], int j, cplx* lds, const cplx* __restrict__ tw)
{
    dft8(u);
    twiddle8(u, tw, 2 * j);
#pragma unroll
    for (int s = 0; s < 8; ++s) lds[256 * s + j] = u[s];
    __syncthreads();
    const int a = j & 31, s1 = j >> 5;
    {
        const cplx* rd = lds + 256 * s1 + a;
#pragma unroll
        for (int b = 0; b < 8; ++b) u[b] = rd[32 * b];
    }
    __syncthreads();
    dft8(u);
    twiddle8(u, tw, 16 * a);
    {
        cplx* wr = lds + a + 36 * s1;
#pragma unroll
        for (int s = 0; s < 8; ++s) wr[288 * s] = u[s];
    }
    __syncthreads();
    const int c = j & 3, Q = j >> 2;
    {
        const cplx* rd = lds + c + 36 * Q;
#pragma unroll
        for (int d = 0; d < 8; ++d) u[d] = rd[4 * d];
    }
    __syncthreads();
    dft8(u);
    twiddle8(u, tw, 128 * c);
    {
        cplx* wr = lds + H2K_P3F * c + Q;
#pragma unroll
        for (int s = 0; s < 8; ++s) wr[64 * s] = u[s];
    }
    __syncthreads();
    const int Cp = j ? 512 - j : 256;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) { u[cc] = lds[H2K_P3F * cc + j]; u[4 + cc] = lds[H2K_P3F * cc + Cp]; }
    dft4(u[0], u[1], u[2], u[3]);
    dft4(u[4], u[5], u[6], u[7]);
}

// In: u[s4] = G[j + 512 s4], u[4 + s4] = G[C' + 512 s4].  Out: u[r] = FFT_2048(G)[j + 256 r].  Same calling rules as fft2048_fwd.
__device__ __forceinline__ void fft2048_tr(cplx (&u)[8], int j, cplx* lds, const cplx* __restrict__ tw)
{
    dft4(u[0], u[1], u[2], u[3]);
    dft4(u[4], u[5], u[6], u[7]);
    const int Cp = j ? 512 - j : 256;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) { lds[H2K_P3T * cc + j] = u[cc]; lds[H2K_P3T * cc + Cp] = u[4 + cc]; }
    __syncthreads();
    const int c = j & 3, Q = j >> 2;
    {
        const cplx* rd = lds + H2K_P3T * c + Q;
#pragma unroll
        for (int s = 0; s < 8; ++s) u[s] = rd[64 * s];
    }
    __syncthreads();
    twiddle8(u, tw, 128 * c);
    dft8(u);
    {
        cplx* wr = lds + c + 36 * Q;
#pragma unroll
        for (int d = 0; d < 8; ++d) wr[4 * d] = u[d];
    }
    __syncthreads();
    const int a = j & 31, s1 = j >> 5;
    {
        const cplx* rd = lds + a + 36 * s1;
#pragma unroll
        for (int s = 0; s < 8; ++s) u[s] = rd[288 * s];
    }
    __syncthreads();
    twiddle8(u, tw, 16 * a);
    dft8(u);
    {
        cplx* wr = lds + 256 * s1 + a;
#pragma unroll
        for (int b = 0; b < 8; ++b) wr[32 * b] = u[b];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) u[s] = lds[256 * s + j];
    twiddle8(u, tw, 2 * j);
    dft8(u);
}

// Real-row untangle.  With z[m] = x[2m] + i x[2m+1], Z = FFT_2048(z), w = tw[k], kp = 2048 - k:
//   X[k]  = h [(Z[k] + conj Z[kp]) - i w (Z[k] - conj Z[kp])],   X[kp] = h [(Z[kp] + conj Z[k]) + i conj(w) (Z[kp] - conj Z[k])]     (h = 1/2).
// The inverse direction uses the same map (h = 1) on the conjugated half spectrum: G[k] = conj(Zs[k]), z = conj(FFT_2048(G)).
__device__ __forceinline__ void untangle2(cplx zk, cplx zp, cplx w, double h, cplx& xk, cplx& xp)
{
    const double Sx = zk.x + zp.x, Sy = zk.y - zp.y, Dx = zk.x - zp.x, Dy = zk.y + zp.y;
    const double wdx = w.x * Dx - w.y * Dy, wdy = w.x * Dy + w.y * Dx;
    xk = make_double2(h * (Sx + wdy), h * (Sy - wdx));
    xp = make_double2(h * (Sx - wdy), -h * (Sy + wdx));
}

#ifndef SFFT_CR
#define SFFT_CR(i, n) (i)
#endif

__device__ __forceinline__ void h2k_store(cplx* p, cplx v)
{
#if H2K_ABLATE & 8
    if (v.x == 1.2345e300) st_stream(p, v);             // (never true: the value is computed, nothing is written)
#else
    st_stream(p, v);
#endif
}

// rows, real -> half complex (N1 = 4096), ONE image row per workgroup, spatial factors fused; the planes [first, first + count) of a
// launch group share their source image (the row is read once).  Workgroup b takes row (b % 8) * rows_per_xcd + b / 8: consecutive
// rows run on one XCD, whose L2 merges the two 64-byte halves (rows 2p, 2p + 1) of a 128-byte line of the 4-column panels.
// The image rows and the column-factor tables must be 16-byte aligned (checked by the launcher).
__global__ void __launch_bounds__(256, H2K_WPS) rows_r2c_4096_h(RowsArgs a, RowGroups grp, cplx* __restrict__ out, int N0, int Nhp, SpecLayout lay,
                                                          const cplx* __restrict__ tw, double scale, int rows_per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N1 = 4096;
    const int j = threadIdx.x;
    const int pfirst = grp.first[blockIdx.y], pcount = grp.count[blockIdx.y];
    const int l = (int)(blockIdx.x & 7) * rows_per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= rows_per_xcd || l >= N0) return;
    const int lm = SFFT_CR(l, 16);                       // (the row whose memory is touched: l itself in the product)
    const cplx* __restrict__ rp = reinterpret_cast<const cplx*>(a.src[pfirst] + (size_t)lm * N1);
    cplx x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = ld_stream(rp + j + 256 * ((H2K_ABLATE & 16) ? 0 : r));          // (x[2m], x[2m + 1]), m = j + 256 r
    const int mnq = grp.mom_nq[blockIdx.y];
    if (mnq > 0) {                          // (workgroup uniform) row moments sum_n x[n] ((n + 1) / N1)^q, q < mnq: see RowGroups
        double acc[ROWMOM_FUSED_MAX];
#pragma unroll
        for (int q = 0; q < ROWMOM_FUSED_MAX; ++q) acc[q] = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double c0 = (double)(2 * (j + 256 * r) + 1) * (1.0 / 4096.0), c1 = (double)(2 * (j + 256 * r) + 2) * (1.0 / 4096.0);
            double p0 = 1.0, p1 = 1.0;
#pragma unroll
            for (int q = 0; q < ROWMOM_FUSED_MAX; ++q) { acc[q] = fma(x[r].x, p0, fma(x[r].y, p1, acc[q])); p0 *= c0; p1 *= c1; }
        }
        double* red = reinterpret_cast<double*>(lds);         // [4 waves][SFFT_MAX_BQ]
#pragma unroll
        for (int q = 0; q < ROWMOM_FUSED_MAX; ++q) {
            if (q < mnq) {
                double v = acc[q];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
                if ((j & 63) == 0) red[(j >> 6) * SFFT_MAX_BQ + q] = v;
            }
        }
        __syncthreads();
        if (j < mnq) grp.mom_out[blockIdx.y][(size_t)l * SFFT_MAX_BQ + j] = (red[j] + red[SFFT_MAX_BQ + j]) + (red[2 * SFFT_MAX_BQ + j] + red[3 * SFFT_MAX_BQ + j]);
        __syncthreads();                    // the transform below reuses this LDS
    }
    const double hs = 0.5 * scale;
    for (int pp = 0; pp < pcount; ++pp) {
        const int plane = pfirst + pp;
        const double cx = a.wx[plane][l];
        const cplx* __restrict__ wy = reinterpret_cast<const cplx*>(a.wy[plane]);
        cplx u[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#if H2K_ABLATE & 2
            const cplx f = make_double2(1.0, 1.0 + 1e-9 * j);
#else
            const cplx f = wy[j + 256 * r];
#endif
#if H2K_RELOAD
            const cplx xv = (pp == 0) ? x[r] : rp[j + 256 * r];      // (planes after the first: the row again, from L2)
#else
            const cplx xv = x[r];
#endif
            u[r] = make_double2(xv.x * (cx * f.x), xv.y * (cx * f.y));
        }
        if (pp > 0) __syncthreads();            // the previous plane's last LDS reads are done
        // (an offset the compiler cannot see through keeps the stage twiddles inside the plane loop: see rows_r2c_4096)
        int zoff;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
        const cplx* __restrict__ twl = tw + zoff;
#if !(H2K_ABLATE & 4)
        fft2048_fwd(u, j, lds, twl);
#endif
        cplx* o = out + (size_t)plane * N0 * Nhp + (size_t)lm * lay.rstride;
        // partner of u[s4] (k = j + 512 s4): u[7 - s4] (k' = 2048 - k); thread 0 holds the self-partnered combos 0 and 256
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = j + 512 * s4;
            const cplx zp = (j == 0) ? u[(4 - s4) & 3] : u[7 - s4];
            cplx xk, xp;
            untangle2(u[s4], zp, twl[k], hs, xk, xp);
            h2k_store(o + lay.col(k), xk);
            h2k_store(o + lay.col(2048 - k), xp);
        }
        if (j == 0) {
#pragma unroll
            for (int s4 = 0; s4 < 2; ++s4) {
                const int k = 256 + 512 * s4;
                cplx xk, xp;
                untangle2(u[4 + s4], u[7 - s4], twl[k], hs, xk, xp);
                h2k_store(o + lay.col(k), xk);
                h2k_store(o + lay.col(2048 - k), xp);
            }
        }
    }
}

// rows, half complex -> real (N1 = 4096), ONE row per workgroup, DIFF epilogue:  DIFF[l][n] = J[l][n] - B(l, n) - IDFT_row(FD[l][.])[n]
// (unnormalised inverse, as rows_c2r_diff: the factor rides on FD).  J, DIFF and the background column tables 16-byte aligned.
template <int NQ>
__global__ void __launch_bounds__(256, 4) rows_c2r_diff_4096_h(const cplx* __restrict__ FD, const double* __restrict__ J,
                                                               const double* __restrict__ bpq, BkgArgs bk, double* __restrict__ DIFF,
                                                               int N0, SpecLayout lay, const cplx* __restrict__ tw, int rows_per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N1 = 4096;
    const int j = threadIdx.x;
    const int l = (int)(blockIdx.x & 7) * rows_per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= rows_per_xcd || l >= N0) return;
    const int lm = SFFT_CR(l, 16);
    const cplx* __restrict__ f = FD + (size_t)lm * lay.rstride;
    const int Cp = j ? 512 - j : 256;
    cplx u[8];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        u[s4] = ld_stream(f + lay.col(j + 512 * s4));
        u[4 + s4] = ld_stream(f + lay.col(Cp + 512 * s4));
    }
    cplx xn = ld_stream(f + lay.col(j == 0 ? 2048 : j));          // (thread 0: the Nyquist column)
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e].y = -u[e].y;                  // Y = conj X
    if (j == 0) { u[0].y = 0.0; xn.y = 0.0; }                      // columns 0 and N1 / 2 of a real row's spectrum are real
    // G[k] = (Y[k] + conj Y[k']) - i w^k (Y[k] - conj Y[k']);  threads > 0 get G[k'] (combo C') from the same call
    {
        cplx g[8];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const cplx yp = (j == 0) ? (s4 == 0 ? xn : u[4 - s4]) : u[7 - s4];
            untangle2(u[s4], yp, tw[j + 512 * s4], 1.0, g[s4], g[7 - s4]);
        }
        if (j == 0) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                cplx dummy;
                untangle2(u[4 + s4], u[7 - s4], tw[256 + 512 * s4], 1.0, g[4 + s4], dummy);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = g[e];
    }
    fft2048_tr(u, j, lds, tw);                  // u[r] = conj(z[m]), m = j + 256 r:  x[2m] = u.x, x[2m + 1] = -u.y
    double cq[NQ];
    bkg_row_coeffs<NQ>(bk, bpq, l, N0, cq);
    const cplx* __restrict__ jr = reinterpret_cast<const cplx*>(J + (size_t)lm * N1);
    cplx* __restrict__ dr = reinterpret_cast<cplx*>(DIFF + (size_t)lm * N1);
    // (loads of a batch first, then its arithmetic and stores: see rows_c2r_diff_4096)
    constexpr int BS = (NQ <= 4) ? 4 : 1;
#pragma unroll
    for (int b0 = 0; b0 < 8; b0 += BS) {
        cplx jv[BS], tb[BS][NQ];
#pragma unroll
        for (int e = 0; e < BS; ++e) {
            const int m = j + 256 * (b0 + e);
            jv[e] = ld_stream(jr + m);
#pragma unroll
            for (int q = 0; q < NQ; ++q) tb[e][q] = reinterpret_cast<const cplx*>(bk.tby + (size_t)min(q, bk.nq - 1) * N1)[m];      // clamped: always valid
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < BS; ++e) {
            const int m = j + 256 * (b0 + e);
            const cplx z = u[b0 + e];
            double B0 = 0.0, B1 = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const double cc = (q < bk.nq) ? cq[q] : 0.0;
                B0 = fma(cc, tb[e][q].x, B0);
                B1 = fma(cc, tb[e][q].y, B1);
            }
            h2k_store(dr + m, make_double2(jv[e].x - B0 - z.x, jv[e].y - B1 + z.y));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

#endif
