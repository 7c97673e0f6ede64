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
// address = (workgroup-uniform pointer) + (32-bit byte offset of the lane): the form the global load / store instructions take as
// scalar base + vector offset.  Written as base[lane_index] the compiler builds a 64-bit address per access instead (the scaled
// index could overflow 32 bits for all it knows), computes all of them ahead of the transform, and spills them.
template <class T> __device__ __forceinline__ T* at_byte(T* base, unsigned off)
{
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off);
}
template <class T> __device__ __forceinline__ const T* at_byte(const T* base, unsigned off)
{
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off);
}

// true in the four waves (threads < 256) that run the radix-24 stage, as a scalar
__device__ __forceinline__ bool f6k_stage3_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) < 4; }

#define F6K_LDS 6528                                 // doubles: 6144 + 6144 / 16 (the padded layout of pad16)

// forward 24-point DFT in registers (24 = 3 * 8: n = 3 b + a, k = d + 8 c), in two steps so that a caller can produce the outputs in
// the order and at the time it wants them (each one is a three-term sum of the intermediate array):
//   dft24_g:  G[a][d] = W24^(a d) * DFT8 over b of x[3 b + a]          dft24_x:  X[d + 8 c] = sum_a G[a][d] W3^(a c)
__device__ __forceinline__ void dft24_g(const cplx (&x)[24], cplx (&G)[3][8])
{
    // W24^m = (C[m], -S[m]), m = a d <= 14
    constexpr double C[15] = { 1.0, 0.96592582628906828675, 0.86602540378443864676, 0.70710678118654752440, 0.5, 0.25881904510252076235, 0.0,
                               -0.25881904510252076235, -0.5, -0.70710678118654752440, -0.86602540378443864676, -0.96592582628906828675, -1.0,
                               -0.96592582628906828675, -0.86602540378443864676 };
    constexpr double S[15] = { 0.0, 0.25881904510252076235, 0.5, 0.70710678118654752440, 0.86602540378443864676, 0.96592582628906828675, 1.0,
                               0.96592582628906828675, 0.86602540378443864676, 0.70710678118654752440, 0.5, 0.25881904510252076235, 0.0,
                               -0.25881904510252076235, -0.5 };
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
}
__device__ __forceinline__ cplx dft24_x(const cplx (&G)[3][8], int d, int c)
{
    const double S3 = 0.86602540378443864676;
    const cplx u0 = G[0][d], u1 = G[1][d], u2 = G[2][d];
    const cplx sm = cadd(u1, u2), df = csub(u1, u2);
    if (c == 0) return cadd(u0, sm);
    const cplx m = make_double2(u0.x - 0.5 * sm.x, u0.y - 0.5 * sm.y);
    return c == 1 ? make_double2(m.x + S3 * df.y, m.y - S3 * df.x) : make_double2(m.x - S3 * df.y, m.y + S3 * df.x);
}

// 6144-point forward FFT, stages 1 and 2.  In: u[r] = x[j + 384 r], j < 384.  Out, threads j < 256 only: xin[jl] = B[jl][s][t] of
// thread q = j = s + 16 t, the input of stage 3: X[j + 256 k3] = DFT24 over jl of W^(jl j) xin[jl]  (twiddle24, then dft24).
// `lds` = F6K_LDS doubles.  Every thread of the 384-thread block must call (barriers inside; threads < 256 still read LDS when it
// returns); tw[k] = W^k, k < 6144.  act = (j < 256) as a WAVE-UNIFORM value (waves 0-3 of the six; see f6k_stage3_wave): tested as a
// lane condition, the values that cross the branch become loop-carried registers of the callers' plane loops.
__device__ __forceinline__ void fft6144_front(cplx (&u)[16], cplx (&xin)[24], int j, bool act, double* lds, const cplx* __restrict__ tw)
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
        if (act) {
#pragma unroll
            for (int r = 0; r < 24; ++r) re[r] = lds[jp + 272 * r];
        }
        __syncthreads();
#pragma unroll
        for (int tx = 0; tx < 16; ++tx) lds[wb + 17 * tx] = u[R16_OUT(tx)].y;
        __syncthreads();
        if (act) {
#pragma unroll
            for (int r = 0; r < 24; ++r) xin[r] = make_double2(re[r], lds[jp + 272 * r]);
        }
    }
}

// x[r] *= W^(r j), r = 1..23, from five table entries (products of at most four factors): 23 loads in flight would cost 92 registers
__device__ __forceinline__ void twiddle24(cplx (&xin)[24], const cplx* __restrict__ tw, int j)
{
    const cplx w1 = tw[j], w2 = tw[2 * j], w4 = tw[4 * j], w8 = tw[8 * j], w16 = tw[16 * j];
    const cplx w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
    xin[1] = cmul(xin[1], w1); xin[2] = cmul(xin[2], w2); xin[3] = cmul(xin[3], w3); xin[4] = cmul(xin[4], w4);
    xin[5] = cmul(xin[5], w5); xin[6] = cmul(xin[6], w6); xin[7] = cmul(xin[7], w7); xin[8] = cmul(xin[8], w8);
    xin[9] = cmul(xin[9], cmul(w8, w1)); xin[10] = cmul(xin[10], cmul(w8, w2)); xin[11] = cmul(xin[11], cmul(w8, w3));
    xin[12] = cmul(xin[12], cmul(w8, w4)); xin[13] = cmul(xin[13], cmul(w8, w5)); xin[14] = cmul(xin[14], cmul(w8, w6));
    xin[15] = cmul(xin[15], cmul(w8, w7)); xin[16] = cmul(xin[16], w16);
    xin[17] = cmul(xin[17], cmul(w16, w1)); xin[18] = cmul(xin[18], cmul(w16, w2)); xin[19] = cmul(xin[19], cmul(w16, w3));
    xin[20] = cmul(xin[20], cmul(w16, w4)); xin[21] = cmul(xin[21], cmul(w16, w5)); xin[22] = cmul(xin[22], cmul(w16, w6));
    xin[23] = cmul(xin[23], cmul(w16, w7));
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
    // addresses as (workgroup-uniform pointer) + (one 32-bit lane offset): the 16 + 24 of them otherwise take two registers each
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + cofs;
    const double* __restrict__ w = g.wx[o];
    const unsigned jo = (unsigned)j * (unsigned)lay.rstride * (unsigned)sizeof(cplx), jw = (unsigned)j * (unsigned)sizeof(double);
    const int lo = g.lo[o], hi = g.hi[o];
    cplx u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int l = j + 384 * r;
        const bool ok = l >= lo && l < hi;
        const cplx* __restrict__ sr = src + (size_t)(384 * r) * rs;
        const double* __restrict__ wr = w + 384 * r;
        const cplx z = ok ? *at_byte(sr, jo) : make_double2(0.0, 0.0);
        const double f = ok ? *at_byte(wr, jw) : 0.0;
        u[r] = make_double2(z.x * f, z.y * f);
    }
    cplx xin[24];
    const bool act = f6k_stage3_wave();
    fft6144_front(u, xin, j, act, lds, tw);
    if (!act) return;
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * plane_sz + cofs;
    twiddle24(xin, tw, j);
    cplx G[3][8];
    dft24_g(xin, G);
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int c = 0; c < 3; ++c) { cplx* __restrict__ dr = dst + (size_t)(256 * (d + 8 * c)) * rs; *at_byte(dr, jo) = dft24_x(G, d, c); }
}

// rows, real -> half complex (N1 = 6144), two image rows per transform, spatial factors fused (see rows_r2c_4096: same arguments and
// workgroup order).  The planes [first, first + count) of a launch group share their source image; the workgroup reads its two rows
// again for every plane (from L2 after the first: keeping them in registers would cost the second workgroup of the CU).
// (one workgroup per CU: held to the 168 registers that would let two share it, the compiler spills -- measured 1.8 ms per six-plane
//  launch at 6144^2 against 1.1 ms for this version and 1.4 ms for the generic kernel)
#ifndef F6K_ROWS_BOUNDS
#define F6K_ROWS_BOUNDS 384
#endif
__global__ void __launch_bounds__(F6K_ROWS_BOUNDS) rows_r2c_6144(RowsArgs a, RowGroups grp, cplx* __restrict__ out, int N0, int Nhp, SpecLayout lay,
                                                     const cplx* __restrict__ tw, double scale, int pairs_per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int N1 = 6144;
    const int j = threadIdx.x;
    const int pfirst = grp.first[blockIdx.y], pcount = grp.count[blockIdx.y];
    const int rp = (int)(blockIdx.x & 7) * pairs_per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= pairs_per_xcd || 2 * rp >= N0) return;
    const int l0 = 2 * rp, l1 = l0 + 1;
    const bool has1 = l1 < N0;
    const double* __restrict__ src = a.src[pfirst];
    const double* r0p = src + (size_t)l0 * N1;
    const double* r1p = src + (size_t)(has1 ? l1 : l0) * N1;       // (read unconditionally, scaled by 0 when there is no second row)
    const double hs = 0.5 * scale;
    const bool act = f6k_stage3_wave();
    for (int pp = 0; pp < pcount; ++pp) {
        const int plane = pfirst + pp;
        const double* __restrict__ wx = a.wx[plane];
        const double* __restrict__ wy = a.wy[plane];
        const double cx0 = wx[l0];
        const double cx1 = has1 ? wx[l1] : 0.0;
        // (a zero the compiler cannot see through: otherwise every plane-invariant index -- image and twiddle addresses, LDS
        //  partner slots, output offsets -- is computed ahead of the plane loop, and spilled)
        int zoff;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
        const int jz = j + zoff;                // this plane's copy of j: every index below is derived from it
        const unsigned jb = (unsigned)jz * (unsigned)sizeof(double);
        cplx u[16];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {        // two batches of 8 x 3 loads: bounds the registers of this phase
            double a0[8], a1[8], cy[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int n0 = 384 * (8 * hb + r);          // (uniform pointer + lane offset jz)
                a0[r] = *at_byte(r0p + n0, jb); a1[r] = *at_byte(r1p + n0, jb); cy[r] = *at_byte(wy + n0, jb);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) u[8 * hb + r] = make_double2(a0[r] * (cx0 * cy[r]), a1[r] * (cx1 * cy[r]));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pp > 0) __syncthreads();            // the previous plane's partner reads are done
        cplx xin[24];
        fft6144_front(u, xin, jz, act, lds, tw);
        // Z = FFT(row0 + i row1) comes out of the last stage in threads < 256 (Z[j + 256 k3], k3 = d + 8 c).  Output m <= 3072 needs its
        // partner Z[N1 - m]: the upper half of Z (k3 >= 12: 3072 complex values, 48 KB) goes to LDS first, then the lower half is
        // produced, paired and stored (nothing but the 24 intermediates of the radix-24 stage stays in registers in between).
        // Z[0] and Z[3072] are their own partners.
        cplx* ldc = reinterpret_cast<cplx*>(lds);
        cplx v[13];
        __syncthreads();                        // every thread has read its stage-3 input
        if (act) {
            cplx G[3][8];
            twiddle24(xin, tw, jz);
            dft24_g(xin, G);
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                v[d] = dft24_x(G, d, 0);
                const cplx x1 = dft24_x(G, d, 1);
                if (d <= 4) v[d + 8] = x1;
                if (d >= 4) ldc[jz + 256 * (d + 8 - 12)] = x1;                           // Z[3072 + i] at slot i
                ldc[jz + 256 * (d + 16 - 12)] = dft24_x(G, d, 2);
            }
        }
        __syncthreads();
        if (act) {
            cplx* o0 = out + (size_t)plane * N0 * Nhp + (size_t)l0 * lay.rstride;
            const size_t cstep = lay.col(256);              // col(jz + 256 k3) = col(jz) + k3 col(256): the panel width divides 256
            const unsigned cj = (unsigned)lay.col(jz) * (unsigned)sizeof(cplx);
#pragma unroll
            for (int k3 = 0; k3 <= 12; ++k3) {
                if (k3 < 12 || j == 0) {
                    const cplx z = v[k3];
                    const cplx* pz = ldc + (3072 - 256 * k3);
                    const cplx zp = (k3 == 12 || (k3 == 0 && jz == 0)) ? z : *(pz - jz);     // Z[N1 - m] (m = 0: slot 3072 does not exist)
                    const cplx zc = make_double2(zp.x, -zp.y);
                    cplx* ob = o0 + (size_t)k3 * cstep;
                    *at_byte(ob, cj) = make_double2(hs * (z.x + zc.x), hs * (z.y + zc.y));
                    if (has1) *at_byte(ob + lay.rstride, cj) = make_double2(hs * (z.y - zc.y), -hs * (z.x - zc.x));
                }
            }
        }
    }
}

#endif
