// fft_r24.hpp -- register-resident fast paths for 6144- and 9216-point axes (16 x 16 x 24 and 16 x 24 x 24).
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_FFT_R24_HPP
#define SFFT_AMD_FFT_R24_HPP

// ================================================================================================
// The generic passes keep a whole sequence in LDS (98 KB at 6144 points, 147 KB at 9216): one workgroup per CU, whose load,
// transform and store phases never overlap, and ten trips of the sequence through LDS.  Here it lives in registers as in
// fft_r16_4096.hpp.  N = 16 * Q * 24 with Q = 16 (6144) or 24 (9216); 24 Q threads; LDS only carries the two exchanges between
// the three stages -- real and imaginary parts one after the other, N * 17 / 16 doubles (52 / 78 KB).
//
// Index algebra (n = j + 24 Q r, k = s + 16 (t + Q k3); W = exp(-2 pi i / N)):
//   stage 1, thread j < 24 Q:              A[j][s]      = sum_{r < 16} x[j + 24 Q r] W16^(r s)
//   stage 2, thread 16 jl + s, jl < 24:    B[jl][s][t]  = sum_{r < Q} W^(24 s r) A[jl + 24 r][s] WQ^(r t)
//   stage 3, thread q = s + 16 t < 16 Q:   X[q + 16 Q k3] = sum_{jl < 24} W^(jl q) B[jl][s][t] W24^(jl k3)
// Stage 2 runs in 384 threads, stage 3 in 16 Q: whole waves in both cases, and the branches around them test the WAVE index as a
// scalar -- tested per lane, the values that cross such a branch take registers in every path.
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


// R24_NT (tuning macro): bit 0 the spectra of cols_fwd_weighted_r24, bit 1 the DIFF rows of rows_c2r_diff_r24, bit 2 the outputs of strided_rader577_r24
// leave with non-temporal stores.  Measured per bit (round 6, profiles/r06_f_ab_r24_nontemporal.txt): bit 0 costs config 3's column pass 3.1 -> 4.7 ms (its
// 32-byte pieces need the L2 to meet their neighbours), bit 1 nothing either way, bit 2 gains config 5 1.7 % (42.9 - 43.0 against 42.1 - 42.3 pairs/s): 4 is the default.
#ifndef R24_NT
#define R24_NT 4
#endif
#if R24_NT & 1
#define R24_ST1(p, v) st_nt(p, v)
#else
#define R24_ST1(p, v) (*(p) = (v))
#endif
#if R24_NT & 2
#define R24_ST2(p, v) st_nt(p, v)
#else
#define R24_ST2(p, v) (*(p) = (v))
#endif
#if R24_NT & 4
#define R24_ST4(p, v) st_nt(p, v)
#else
#define R24_ST4(p, v) (*(p) = (v))
#endif
template <int Q> struct R24 {
    static constexpr int N = 384 * Q;                // 6144, 9216
    static constexpr int NT = 24 * Q;                // threads: 384, 576
    static constexpr int NQ = 16 * Q;                // stage-3 threads and output stride: 256, 384
    static constexpr int P2 = NQ + Q;                // padded stride of the stage-3 reads: pad16(q + NQ r) = q + (q >> 4) + P2 r
    static constexpr int LDS = N + N / 16;           // doubles (the padded layout of pad16)
    __device__ static __forceinline__ bool stage2_wave() { return Q == 16 || __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) < 6; }
    __device__ static __forceinline__ bool stage3_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) < NQ / 64; }
};

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


// N-point forward FFT, stages 1 and 2.  In: u[r] = x[j + NT r], j < NT.  Out, threads j < NQ only: xin[jl] = B[jl][s][t] of thread
// q = j = s + 16 t, the input of stage 3: X[j + NQ k3] = DFT24 over jl of W^(jl j) xin[jl]  (twiddle24, then dft24_g / dft24_x).
// `lds` = R24<Q>::LDS doubles.  Every thread of the NT-thread block must call (barriers inside; the stage-3 threads still read LDS
// when it returns); tw[k] = W^k, k < N; act2 / act3 = R24<Q>::stage2_wave() / stage3_wave().
template <int Q>
__device__ __forceinline__ void fft_r24_front(cplx (&u)[16], cplx (&xin)[24], int j, bool act2, bool act3, double* lds, const cplx* __restrict__ tw)
{
    typedef R24<Q> F;
    const int jp = j + (j >> 4);                     // pad16(j + 384 r) = jp + 408 r,  pad16(j + NQ r) = jp + P2 r
    dft16(u);
    // exchange 1: A[j][s] sits at element 16 j + s; the stage-2 thread j = 16 jl + s takes elements j + 384 r, r < Q
    if constexpr (Q == 16) {
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
        const int s = j & 15, jl = j >> 4;
        const int wb = F::P2 * jl + s;               // B[jl][s][t] goes to element NQ jl + s + 16 t: padded, wb + 17 t
        twiddle16(u, tw, 24 * s);
        dft16(u);
        // exchange 2: the stage-3 thread q takes elements q + NQ r, r < 24
        double r3[24];
#pragma unroll
        for (int tx = 0; tx < 16; ++tx) lds[wb + 17 * tx] = u[R16_OUT(tx)].x;
        __syncthreads();
        if (act3) {
#pragma unroll
            for (int r = 0; r < 24; ++r) r3[r] = lds[jp + F::P2 * r];
        }
        __syncthreads();
#pragma unroll
        for (int tx = 0; tx < 16; ++tx) lds[wb + 17 * tx] = u[R16_OUT(tx)].y;
        __syncthreads();
        if (act3) {
#pragma unroll
            for (int r = 0; r < 24; ++r) xin[r] = make_double2(r3[r], lds[jp + F::P2 * r]);
        }
    } else {
        cplx y[24];
        {
            double re[24];
#pragma unroll
            for (int sx = 0; sx < 16; ++sx) lds[17 * j + sx] = u[R16_OUT(sx)].x;
            __syncthreads();
            if (act2) {
#pragma unroll
                for (int r = 0; r < 24; ++r) re[r] = lds[jp + 408 * r];
            }
            __syncthreads();
#pragma unroll
            for (int sx = 0; sx < 16; ++sx) lds[17 * j + sx] = u[R16_OUT(sx)].y;
            __syncthreads();
            if (act2) {
#pragma unroll
                for (int r = 0; r < 24; ++r) y[r] = make_double2(re[r], lds[jp + 408 * r]);
            }
            __syncthreads();
        }
        const int s = j & 15, jl = j >> 4;
        const int wb = F::P2 * jl + s;
        cplx G[3][8];
        if (act2) {
            twiddle24(y, tw, 24 * s);
            dft24_g(y, G);
        }
        // exchange 2 (t = d + 8 c)
        double r3[24];
        if (act2) {
#pragma unroll
            for (int d = 0; d < 8; ++d)
#pragma unroll
                for (int c = 0; c < 3; ++c) lds[wb + 17 * (d + 8 * c)] = dft24_x(G, d, c).x;
        }
        __syncthreads();
        if (act3) {
#pragma unroll
            for (int r = 0; r < 24; ++r) r3[r] = lds[jp + F::P2 * r];
        }
        __syncthreads();
        if (act2) {
#pragma unroll
            for (int d = 0; d < 8; ++d)
#pragma unroll
                for (int c = 0; c < 3; ++c) lds[wb + 17 * (d + 8 * c)] = dft24_x(G, d, c).y;
        }
        __syncthreads();
        if (act3) {
#pragma unroll
            for (int r = 0; r < 24; ++r) xin[r] = make_double2(r3[r], lds[jp + F::P2 * r]);
        }
    }
}

// Forward column pass of the weighted planes for N0 = 384 Q (see cols_fwd_weighted: same arguments, same XCD-aware order -- on one
// XCD the eight columns that share 128-byte lines, then the next output of the same column group).
// History of the one-column form at config 3 (docs/LOG.md, round 4): 13.8 GB written for 7.55 GB of spectra (the four 16-byte writers
// of a 64-byte piece are sibling workgroups whose lines leave L2 incomplete) -- yet a bounded meeting of the siblings before their
// store phase, which brought the writes to 7.7 GB, made the kernel SLOWER (4.31 -> 4.52 ms), and so did two columns per workgroup in
// separate waves (4.31 vs 4.10).  What it waits for is the rate of its 16-byte requests.
// NC = 2 (6144 points): two panel neighbours per workgroup in NEIGHBOURING LANES (lane parity = column, team thread = lane >> 1), so
// that every load / store instruction moves 32 contiguous bytes per lane pair: half the requests of the one-column form, which is
// bound by the RATE of its 16-byte requests (64 distinct 64-byte pieces per wave instruction), not by bytes.  Each team has its own
// LDS region, 16 doubles out of phase with the other's so that the lane pairs of a half-wave fall into complementary banks.
template <int Q, int NC = 1>
__global__ void __launch_bounds__(24 * Q * NC) cols_fwd_weighted_r24(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int ncols,
                                                                     int Nhp, SpecLayout lay, SpecLayout lay_out, const cplx* __restrict__ tw)
{
    // lay_out: layout of the spectra (= lay, or -- NC = 2, round 6 -- 2-column panels: the lane pairs of a wave then store 1 KB contiguous instead of
    // 32-byte pieces of 64-byte rows)
    typedef R24<Q> F;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int N0 = F::N;
    const int c = NC > 1 ? (int)(threadIdx.x & (NC - 1)) : 0;
    const int j = (int)threadIdx.x / NC;
    double* lds = reinterpret_cast<double*>(smem_raw) + c * (F::LDS + 16);
    const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    constexpr int GW = 8 / NC;                                   // workgroups per 8-column group
    const int gq = (t % GW) * NC + c, o = (t / GW) % g.nout, tg = (t / GW) / g.nout;
    const int col0 = (tg * 8 + xcd) * 8 + gq;
    if (col0 - c >= ncols) return;     // workgroup-uniform (col0 - c is the team-0 column; no barrier reached yet): every column of the workgroup is past the end
    const int col = NC == 1 ? col0 : min(col0, ncols - 1);       // (a column past the end repeats the last one and is not stored: the teams share barriers)
    const size_t plane_sz = (size_t)N0 * Nhp, cofs = lay.col(col), rs = (size_t)lay.rstride;
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + cofs;
    const double* __restrict__ w = g.wx[o];
    const unsigned jo = (unsigned)j * (unsigned)lay.rstride * (unsigned)sizeof(cplx), jw = (unsigned)j * (unsigned)sizeof(double);
    const int lo = g.lo[o], hi = g.hi[o];
    cplx u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int l = j + F::NT * r;
        const bool ok = l >= lo && l < hi;
        const cplx* __restrict__ sr = src + (size_t)(F::NT * r) * rs;
        const double* __restrict__ wr = w + F::NT * r;
        const cplx z = ok ? *at_byte(sr, jo) : make_double2(0.0, 0.0);
        const double f = ok ? *at_byte(wr, jw) : 0.0;
        u[r] = make_double2(z.x * f, z.y * f);
    }
    cplx xin[24];
    const int wv = __builtin_amdgcn_readfirstlane(j >> 6);      // wave of the team's thread numbering (uniform: a wave holds 64 / NC consecutive j)
    const bool act = wv < F::NQ / 64, act2 = Q == 16 || wv < 6;
    fft_r24_front<Q>(u, xin, j, act2, act, lds, tw);
    if (!act || col0 >= ncols) return;
    const size_t rso = (size_t)lay_out.rstride;
    const unsigned joo = (unsigned)j * (unsigned)lay_out.rstride * (unsigned)sizeof(cplx);
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * plane_sz + lay_out.col(col);
    twiddle24(xin, tw, j);
    cplx G[3][8];
    dft24_g(xin, G);
    if (NC == 2 && lay_out.mask == 1) {     // (launch uniform) 2-column panels: whole contiguous kilobytes per wave, non-temporal (2.89 -> 2.69 ms at config 3)
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) { cplx* __restrict__ dr = dst + (size_t)(F::NQ * (d + 8 * cc)) * rso; st_nt(at_byte(dr, joo), dft24_x(G, d, cc)); }
        return;
    }
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) { cplx* __restrict__ dr = dst + (size_t)(F::NQ * (d + 8 * cc)) * rso; R24_ST1(at_byte(dr, joo), dft24_x(G, d, cc)); }
}

// rows, real -> half complex (N1 = 384 Q), two image rows per transform, spatial factors fused (see rows_r2c_4096 for the arguments).
// One workgroup per (row pair, plane); the plane index runs fastest in an XCD's share of the grid, so the planes of one image read
// the row pair from that XCD's L2 after the first.  (A plane loop inside the workgroup, as in rows_r2c_4096, costs this kernel 80
// more registers.)
template <int Q>
__global__ void __launch_bounds__(24 * Q, 3) rows_r2c_r24(RowsArgs a, int nplanes, cplx* __restrict__ out, int N0, int Nhp, SpecLayout lay,
                                                          const cplx* __restrict__ tw, double scale, int pairs_per_xcd)
{
    typedef R24<Q> F;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    constexpr int N1 = F::N, NH = F::N / 2, NQ = F::NQ;
    const int j = threadIdx.x;
    const int item = (int)(blockIdx.x >> 3);
    const int plane = item % nplanes;
    const int rp = (int)(blockIdx.x & 7) * pairs_per_xcd + item / nplanes;
    if (item / nplanes >= pairs_per_xcd || 2 * rp >= N0) return;
    const int l0 = 2 * rp, l1 = l0 + 1;
    const bool has1 = l1 < N0;
    const double* __restrict__ src = a.src[plane];
    const double* r0p = src + (size_t)l0 * N1;
    const double* r1p = src + (size_t)(has1 ? l1 : l0) * N1;       // (read unconditionally, scaled by 0 when there is no second row)
    const double hs = 0.5 * scale;
    const bool act = F::stage3_wave();
    const double* __restrict__ wx = a.wx[plane];
    const double* __restrict__ wy = a.wy[plane];
    const double cx0 = wx[l0];
    const double cx1 = has1 ? wx[l1] : 0.0;
    const unsigned jb = (unsigned)j * (unsigned)sizeof(double);
    cplx u[16];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {            // two batches of 8 x 3 loads: bounds the registers of this phase
        double a0[8], a1[8], cy[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n0 = F::NT * (8 * hb + r);            // (uniform pointer + lane offset)
            a0[r] = *at_byte(r0p + n0, jb); a1[r] = *at_byte(r1p + n0, jb); cy[r] = *at_byte(wy + n0, jb);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) u[8 * hb + r] = make_double2(a0[r] * (cx0 * cy[r]), a1[r] * (cx1 * cy[r]));
        __builtin_amdgcn_sched_barrier(0);
    }
    cplx xin[24];
    fft_r24_front<Q>(u, xin, j, F::stage2_wave(), act, lds, tw);
    // Z = FFT(row0 + i row1) comes out of the last stage in the stage-3 threads (Z[j + NQ k3], k3 = d + 8 c).  Output m <= N1 / 2 needs
    // its partner Z[N1 - m]: the upper half of Z (k3 >= 12: N1 / 2 complex values) goes to LDS as it is produced; Z[0] and Z[N1 / 2]
    // are their own partners.
    cplx* ldc = reinterpret_cast<cplx*>(lds);
    cplx v[13];
    __syncthreads();                            // every thread has read its stage-3 input
    if (act) {
        cplx G[3][8];
        twiddle24(xin, tw, j);
        dft24_g(xin, G);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            v[d] = dft24_x(G, d, 0);
            const cplx x1 = dft24_x(G, d, 1);
            if (d <= 4) v[d + 8] = x1;
            if (d >= 4) ldc[j + NQ * (d + 8 - 12)] = x1;                                 // Z[N1 / 2 + i] at slot i
            ldc[j + NQ * (d + 16 - 12)] = dft24_x(G, d, 2);
        }
    }
    __syncthreads();
    if (act) {
        cplx* o0 = out + (size_t)plane * N0 * Nhp + (size_t)l0 * lay.rstride;
        const size_t cstep = lay.col(NQ);                   // col(j + NQ k3) = col(j) + k3 col(NQ): the panel width divides NQ
        const unsigned cj = (unsigned)lay.col(j) * (unsigned)sizeof(cplx);
#pragma unroll
        for (int k3 = 0; k3 <= 12; ++k3) {
            if (k3 < 12 || j == 0) {
                const cplx z = v[k3];
                const cplx* pz = ldc + (NH - NQ * k3);
                const cplx zp = (k3 == 12 || (k3 == 0 && j == 0)) ? z : *(pz - j);         // Z[N1 - m] (m = 0: slot N1 / 2 does not exist)
                const cplx zc = make_double2(zp.x, -zp.y);
                cplx* ob = o0 + (size_t)k3 * cstep;
                *at_byte(ob, cj) = make_double2(hs * (z.x + zc.x), hs * (z.y + zc.y));
                if (has1) *at_byte(ob + lay.rstride, cj) = make_double2(hs * (z.y - zc.y), -hs * (z.x - zc.x));
            }
        }
    }
}

// rows, half complex -> real (N1 = 384 Q), two rows per transform, DIFF epilogue: the register-resident twin of rows_c2r_diff (same
// arguments; see rows_c2r_diff_4096 for the packing).  The generic pass keeps the 6144- / 9216-point sequence in LDS (one workgroup
// per CU, ten trips through LDS): 0.49 ms of a config-3 pair, 1.16 ms of a config-5 pair (1.8 TB/s).  Z = X0 + i X1 is conjugated on
// input so that the forward transform acts as the inverse; thread j loads the elements m = j + NT r, the mirrored half of the
// spectrum (m > N1 / 2) from column N1 - m with the sign of its imaginary part flipped; the stage-3 threads hold the 24 results
// n = j + NQ k3 of both rows and finish them in batches of EB (all J / background-table loads of a batch before its stores).
// LB / EB: elements per batch of loads / results per batch of the epilogue (9216 points: 576 threads = three waves on one SIMD,
// 168 registers each).
template <int Q, int NQB>
__global__ void __launch_bounds__(24 * Q) rows_c2r_diff_r24(const cplx* __restrict__ FD, const double* __restrict__ J, const double* __restrict__ bpq,
                                                            BkgArgs bk, double* __restrict__ DIFF, int N0, SpecLayout lay, const cplx* __restrict__ tw)
{
    typedef R24<Q> F;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    constexpr int N1 = F::N, NQ = F::NQ;
    constexpr int LB = (Q == 24) ? 4 : 8, EB = (Q == 24) ? 2 : 4;
    const int j = threadIdx.x;
    const int l0 = 2 * blockIdx.x, l1 = l0 + 1;
    const bool has1 = l1 < N0;
    const cplx* __restrict__ f0 = FD + (size_t)l0 * lay.rstride;
    const cplx* __restrict__ f1 = FD + (size_t)(has1 ? l1 : l0) * lay.rstride;
    const double h1 = has1 ? 1.0 : 0.0;
    // (the background coefficients of the two rows first, while nothing else is live)
    double c0[NQB], c1[NQB];
    bkg_row_coeffs<NQB>(bk, bpq, l0, N0, c0);
    bkg_row_coeffs<NQB>(bk, bpq, has1 ? l1 : l0, N0, c1);
#pragma unroll
    for (int q = 0; q < NQB; ++q) {             // the row's coefficients are the same in every lane: keep them in scalar registers
        const bool on = q < bk.nq;
        const double v0 = on ? c0[q] : 0.0, v1 = on ? c1[q] : 0.0;
        c0[q] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v0)), __builtin_amdgcn_readfirstlane(__double2loint(v0)));
        c1[q] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v1)), __builtin_amdgcn_readfirstlane(__double2loint(v1)));
    }
    // N1 / 2 = 8 NT: the elements r < 8 (m = j + NT r < N1 / 2) come straight from column m, the elements r >= 8 from column N1 - m with
    // the imaginary part's sign flipped -- known per r at compile time; m = 0 and m = N1 / 2 (thread 0, r = 0 and r = 8) are real
    cplx u[16];
    const double keep = (j == 0) ? 0.0 : 1.0;
#pragma unroll
    for (int hb = 0; hb < 16 / LB; ++hb) {
        cplx a0[LB], a1[LB];
#pragma unroll
        for (int r = 0; r < LB; ++r) {
            const int rr = LB * hb + r;
            const int mm = (rr < 8) ? j + F::NT * rr : N1 - F::NT * rr - j;
            const unsigned mo = (unsigned)lay.col(mm) * (unsigned)sizeof(cplx);
            a0[r] = *at_byte(f0, mo); a1[r] = *at_byte(f1, mo);
        }
#pragma unroll
        for (int r = 0; r < LB; ++r) {
            const int rr = LB * hb + r;
            cplx x0 = a0[r], x1 = make_double2(a1[r].x * h1, a1[r].y * h1);
            if (rr == 0 || rr == 8) { x0.y *= keep; x1.y *= keep; }
            if (rr >= 8) { x0.y = -x0.y; x1.y = -x1.y; }
            u[rr] = make_double2(x0.x - x1.y, -(x0.y + x1.x));      // conj(X0 + i X1)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    cplx xin[24];
    const bool act = F::stage3_wave();
    fft_r24_front<Q>(u, xin, j, F::stage2_wave(), act, lds, tw);
    if (!act) return;
    twiddle24(xin, tw, j);
    cplx G[3][8];
    dft24_g(xin, G);
    __builtin_amdgcn_sched_barrier(0);          // (the epilogue's loads issued ahead of this point are spilled at 9216 points)
    const double* __restrict__ j0 = J + (size_t)l0 * N1;
    const double* __restrict__ j1 = J + (size_t)(has1 ? l1 : l0) * N1;
    double* __restrict__ d0 = DIFF + (size_t)l0 * N1;
    double* __restrict__ d1 = DIFF + (size_t)(has1 ? l1 : l0) * N1;
    const unsigned jb = (unsigned)j * (unsigned)sizeof(double);
#pragma unroll
    for (int cb = 0; cb < 24 / EB; ++cb) {
        const int c = cb / (8 / EB), dq = EB * (cb % (8 / EB));
        double jv0[EB], jv1[EB], tb[EB][NQB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int n0 = NQ * (dq + e + 8 * c);
            jv0[e] = *at_byte(j0 + n0, jb);
            jv1[e] = *at_byte(j1 + n0, jb);
#pragma unroll
            for (int q = 0; q < NQB; ++q) tb[e][q] = *at_byte(bk.tby + (size_t)min(q, bk.nq - 1) * N1 + n0, jb);      // clamped: always valid
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int n0 = NQ * (dq + e + 8 * c);
            const cplx z = dft24_x(G, dq + e, c);    // conj(result): row 0 = z.x, row 1 = -z.y
            double B0 = 0.0, B1 = 0.0;
#pragma unroll
            for (int q = 0; q < NQB; ++q) {
                B0 = fma(c0[q], tb[e][q], B0);
                B1 = fma(c1[q], tb[e][q], B1);
            }
            R24_ST2(at_byte(d0 + n0, jb), jv0[e] - B0 - z.x);
            if (has1) R24_ST2(at_byte(d1 + n0, jb), jv1[e] - B1 + z.y);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ================================================================================================
// The 577-point sub-transform of a four-step column axis (9232 = 16 x 577, config 5) by Rader's algorithm, in registers:
// the two 576-point transforms of the cyclic convolution are 24 x 24, 24 threads per sequence with 24 points each, and every stage is
// a dft24 on a thread's own registers -- LDS carries one exchange per transform (real and imaginary parts one after the other) where
// strided_rader577 makes six passes of three stages each over an LDS-resident tile.  Eight sequences (adjacent columns) per workgroup
// of 192 threads, thread = 8 i + sequence: every load and store instruction moves whole 128-byte rows of the eight columns, gathered /
// scattered in Rader order straight from / to global memory (gin[r] = g^r, gout[q] = g^-q behind rin / rout).
//   a[r] = x[g^r];  A = FFT576(a);  c = conj(A .* bf);  c[0] += conj(x[0]);  C = FFT576(c);  X[g^-q] = conj(C[q]);  X[0] = x[0] + A[0]
// (bf = FFT576(b) / 576, b[q] = W577^(g^-q), bf[0] = -1 / 576: see lds_rader577).  Same PassDesc as strided_rader577 (mode 2, no weights).
// ================================================================================================
#define RDR_SEQ 8
#define RDR_NT (24 * RDR_SEQ)
#define RDR_SS 616                                   // doubles per sequence: 24 rows of 25 (padded), = 8 mod 32 so that the eight sequences spread over the banks

// one 576-point transform step: t[s] (this thread's stage-1 outputs, s < 24, already twiddled) -> z[r] = the value of thread r, slot of this thread
__device__ __forceinline__ void rdr_exchange(const cplx (&t)[24], cplx (&z)[24], double* sb, int i)
{
    double zr[24];
#pragma unroll
    for (int sx = 0; sx < 24; ++sx) sb[25 * sx + i] = t[sx].x;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 24; ++r) zr[r] = sb[25 * i + r];
    __syncthreads();
#pragma unroll
    for (int sx = 0; sx < 24; ++sx) sb[25 * sx + i] = t[sx].y;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 24; ++r) z[r] = make_double2(zr[r], sb[25 * i + r]);
    __syncthreads();
}

// in: y[r] = x[i + 24 r] of a 576-point sequence (thread i of its 24).  out: F[r] = X[i + 24 r].
__device__ __forceinline__ void rdr_fft576(cplx (&y)[24], cplx (&F)[24], double* sb, int i, const cplx* __restrict__ tw)
{
    cplx G[3][8], t[24], z[24];
    dft24_g(y, G);
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int c = 0; c < 3; ++c) t[d + 8 * c] = dft24_x(G, d, c);
    twiddle24(t, tw, i);                             // W576^(i s)
    rdr_exchange(t, z, sb, i);
    dft24_g(z, G);
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int c = 0; c < 3; ++c) F[d + 8 * c] = dft24_x(G, d, c);
}

__global__ void __launch_bounds__(RDR_NT, 3) strided_rader577_r24(const cplx* __restrict__ in, cplx* __restrict__ out, PassDesc d, AxisDev ax)
{
    __shared__ double lds[RDR_SEQ * RDR_SS];
    constexpr int N = 577;
    const int tid = threadIdx.x, seq = tid & (RDR_SEQ - 1), i = tid >> 3;
    const int j = (int)blockIdx.y;
    const int line = (int)blockIdx.x * RDR_SEQ + seq;
    const bool lok = line < d.nlines;
    const int* __restrict__ gin = ax.rin + N;
    const int* __restrict__ gout = ax.rout + N;
    // (uniform base + 32-bit lane offset, see at_byte: a plane is < 4 GB)
    const cplx* __restrict__ pin = in + (long long)blockIdx.x * RDR_SEQ * d.lst_in + (long long)j * d.js_in;
    const unsigned lin = (unsigned)((lok ? seq : d.nlines - 1 - (int)blockIdx.x * RDR_SEQ) * (int)d.lst_in) * (unsigned)sizeof(cplx);
    const unsigned ein = (unsigned)d.es_in * (unsigned)sizeof(cplx);
    double* sb = lds + seq * RDR_SS;
    cplx y[24], F[24];
#pragma unroll
    for (int r = 0; r < 24; ++r) y[r] = *at_byte(pin, lin + (unsigned)gin[i + 24 * r] * ein);
    cplx x0 = *at_byte(pin, lin);
    if (d.conj_in) {
#pragma unroll
        for (int r = 0; r < 24; ++r) y[r].y = -y[r].y;
        x0.y = -x0.y;
    }
    rdr_fft576(y, F, sb, i, ax.tw);                  // F[k] = A[i + 24 k]
    const cplx A0 = F[0];                            // (thread i = 0: A[0] = sum of a)
#pragma unroll
    for (int k = 0; k < 24; ++k) y[k] = cconj(cmul(F[k], ax.bf[i + 24 * k]));
    if (i == 0) { y[0].x += x0.x; y[0].y -= x0.y; }
    rdr_fft576(y, F, sb, i, ax.tw);                  // conj(F[k]) = x[0] + (a (*) b)[i + 24 k] = X[g^-(i + 24 k)]
    if (!lok) return;
    cplx* __restrict__ pout = out + (long long)blockIdx.x * RDR_SEQ * d.lst_out + (long long)j * d.js_out;
    const unsigned lout = (unsigned)(seq * (int)d.lst_out) * (unsigned)sizeof(cplx), eout = (unsigned)d.es_out * (unsigned)sizeof(cplx);
    const double sy = d.conj_out ? d.scale : -d.scale;          // (the conjugation of the second transform, and the caller's)
#pragma unroll
    for (int k = 0; k < 24; ++k) R24_ST4(at_byte(pout, lout + (unsigned)gout[i + 24 * k] * eout), make_double2(F[k].x * d.scale, F[k].y * sy));
    if (i == 0) {
        const cplx X0 = cadd(x0, A0);
        *at_byte(pout, lout) = make_double2(X0.x * d.scale, d.conj_out ? -X0.y * d.scale : X0.y * d.scale);
    }
}

#endif
