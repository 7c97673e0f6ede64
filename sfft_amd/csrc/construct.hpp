// construct.hpp -- Construct_FDIFF and the small spectrum-arithmetic kernels of the FFT utilities.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_CONSTRUCT_HPP
#define SFFT_AMD_CONSTRUCT_HPP

// ------------------------------------------------------------------------------------------------
// Subtraction: kernel transfer function + Construct_FDIFF (SFFTConfigure.py:737-809)
//   FD[l][m] = sum_ij FI_ij[l][m] * SCALE * ( sum_ab a_ijab W0^(l a) W1^(m b)  -  Soff[ij] ),   Soff[ij] = sum_{ab != centre} a_ijab
// The sum over a is done once per image ROW into a small table (kernel_rtab),
//   R_b[ij][l] = sum_a a_ijab W0^(l a),        Kf[ij][l][m] = SCALE * ( sum_b R_b W1^(m b) - Soff ),
// stored as  R_0 - Soff  and, for b = 1..w,  P_b = R_b + R_-b,  M_b = R_b - R_-b,  because with |W1| = 1
//   R_b W1^(mb) + R_-b W1^(-mb) = (wx Px - wy My) + i (wx Py + wy Mx),     wx + i wy = W1^(m b):
// four real FMAs per (ij, b, element).  In construct_fd a wave owns 64 columns and walks down the rows: its W1^(m b) live in
// registers, the table row is wave-uniform (scalar loads), and the only vector memory traffic is the one streaming read
// of the FI planes.  (The first version kept a [ij][a][m] table and re-read it per 8 rows: 3x the plane traffic in L2.)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kernel_rtab(const double* __restrict__ sol, cplx* __restrict__ Rtab, int Fij, int L0, int L1,
                                                   int w0, int w1, int N0, int RS, const cplx* __restrict__ root0, int skip_centre)
{
    // one thread per (ij, l, b >= 0); RS = row stride of the table in complex elements (1 + 2 * padded w1)
    // skip_centre: separately varying scaling -- the centre coefficient a_ij00 does not multiply kernel plane ij
    // (BSplineSFFT.py:2489-2497); its term is applied in real space by scaling_term()
    const int b = blockIdx.y;                     // 0 .. w1
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int ij = blockIdx.z;
    if (l >= N0) return;
    const int Fab = L0 * L1;
    const double* a_ij = sol + (size_t)ij * Fab;
    double rpx = 0.0, rpy = 0.0, rmx = 0.0, rmy = 0.0;      // R_b, R_-b
    for (int aa = 0; aa < L0; ++aa) {
        long long q = ((long long)l * (aa - w0)) % N0; if (q < 0) q += N0;
        const cplx w = root0[q];
        double cp = a_ij[aa * L1 + (w1 + b)], cm = a_ij[aa * L1 + (w1 - b)];
        if (skip_centre && b == 0 && aa == w0) { cp = 0.0; cm = 0.0; }
        rpx = fma(cp, w.x, rpx); rpy = fma(cp, w.y, rpy);
        rmx = fma(cm, w.x, rmx); rmy = fma(cm, w.y, rmy);
    }
    cplx* row = Rtab + ((size_t)ij * N0 + l) * RS;
    if (b == 0) {
        double so = 0.0;
        const int cen = w0 * L1 + w1;
        for (int ab = 0; ab < Fab; ++ab) if (ab != cen) so += a_ij[ab];
        row[0] = make_double2(rpx - so, rpy);
    } else {
        row[2 * b - 1] = make_double2(rpx + rmx, rpy + rmy);
        row[2 * b] = make_double2(rpx - rmx, rpy - rmy);
    }
}

// W = compile-time bound on w1 (table rows are zero padded to it); U rows and G planes in flight per wave
template <int W, int U, int G>
__global__ void __launch_bounds__(64) construct_fd(const cplx* __restrict__ FI, cplx* __restrict__ FD, const cplx* __restrict__ Rtab,
                                                   const cplx* __restrict__ root1, int N0, int N1, int Nh, int Nhp, SpecLayout lay,
                                                   int Fij, int rows_per_wave, double scale)
{
    constexpr int RS = 1 + 2 * W;
    const int lane = threadIdx.x;
    const int m = blockIdx.x * 64 + lane;
    const bool active = m < Nh;
    const int mc = active ? m : 0;
    double wx[W], wy[W];
#pragma unroll
    for (int b = 1; b <= W; ++b) {
        const cplx t = root1[(int)(((long long)mc * b) % N1)];
        wx[b - 1] = t.x; wy[b - 1] = t.y;
    }
    const size_t plane_sz = (size_t)N0 * Nhp;
    const size_t mo = lay.col(mc), rs = (size_t)lay.rstride;
    const int lb = blockIdx.y * rows_per_wave;
    const int le = min(N0, lb + rows_per_wave);
    for (int l = lb; l < le; l += U) {
        cplx acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = make_double2(0.0, 0.0);
        for (int ij0 = 0; ij0 < Fij; ij0 += G) {
            cplx fi[G][U];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int ij = min(ij0 + g, Fij - 1);
#pragma unroll
                for (int u = 0; u < U; ++u) fi[g][u] = FI[(size_t)ij * plane_sz + (size_t)min(l + u, N0 - 1) * rs + mo];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (ij0 + g < Fij) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const cplx* __restrict__ row = Rtab + ((size_t)(ij0 + g) * N0 + min(l + u, N0 - 1)) * RS;   // wave-uniform
                        double kx = row[0].x, ky = row[0].y;
#pragma unroll
                        for (int b = 0; b < W; ++b) {
                            const cplx P = row[2 * b + 1], M = row[2 * b + 2];
                            kx = fma(wx[b], P.x, fma(-wy[b], M.y, kx));
                            ky = fma(wx[b], P.y, fma(wy[b], M.x, ky));
                        }
                        acc[u].x = fma(fi[g][u].x, kx, fma(-fi[g][u].y, ky, acc[u].x));
                        acc[u].y = fma(fi[g][u].x, ky, fma(fi[g][u].y, kx, acc[u].y));
                    }
                }
            }
        }
        if (active) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (l + u < le) FD[(size_t)(l + u) * rs + mo] = make_double2(scale * acc[u].x, scale * acc[u].y);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Mixed-domain apply for polynomial kernels on the staged fast path.  With S_j = row-DFT(I * cy^j) (the stage planes of the row
// pass) and FI_ij = column-DFT(cx^i * S_j), the inverse column DFT of Construct_FDIFF's sum can be taken analytically:
//   sum_l FI_ij[l][m] W0^(l a) e^(+2 pi i l x / N0) = N0 * cx^i[x - a] * S_j[x - a][m]
//   => D[x][m] = N0 * SCALE * sum_ij ( sum_a C_ij[a][m] cx^i[x-a] S_j[x-a][m]  -  Soff_ij cx^i[x] S_j[x][m] ),
//      C_ij[a][m] = sum_b a_ijab W1^(m b),
// a (2w+1)-tap convolution ALONG THE COLUMNS of the stage planes with per-(tap, spectrum column) complex weights.  D is what
// the inverse column pass would have produced, so the apply pass needs no column transform at all: row pass (DK + 1
// planes), this kernel, inverse row pass.  (Replaces cols_fwd_weighted + construct_fd + the inverse cols_c2c there.)
//
// kernel_ctab_mixed: C'[t][a + W][m] = N0 * SCALE * (C_t[a][m] - [a == 0] Soff_t), zero for |a| > w (W = padded half width).
// vconv_mixed: a wave is 16 spectrum columns x 4 independent row streams; the workgroup's 16 streams share the C' slice of
// their 16 columns in LDS.  A stream walks down KS * L source rows (L = 2W + 1) and scatters each into the L output rows it
// touches; the L accumulators are a window that slides down one row per source row.  Output row x is complete right
// after source row x + W.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kernel_ctab_mixed(const double* __restrict__ sol, cplx* __restrict__ Ctab, int L0, int L1, int w0, int w1,
                                                         int W, int Nh, int Nhp, int N1, const cplx* __restrict__ root1, double f)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int LT = 2 * W + 1;
    const int t = blockIdx.y / LT, a = blockIdx.y % LT - W;
    const int Fab = L0 * L1;
    // centre row: S_off = sum of the term's off-centre coefficients, the same for every column -- reduced once per workgroup
    // (every thread used to sum all Fab entries itself)
    __shared__ double so_part[4];
    double so = 0.0;
    if (a == 0) {                                    // (uniform: blockIdx.y)
        const int cen = w0 * L1 + w1;
        const double* at = sol + (size_t)t * Fab;
        double v = 0.0;
        for (int ab = threadIdx.x; ab < Fab; ab += 256) v += (ab != cen) ? at[ab] : 0.0;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0) so_part[threadIdx.x >> 6] = v;
        __syncthreads();
        so = so_part[0] + so_part[1] + so_part[2] + so_part[3];
    }
    if (m >= Nhp) return;
    cplx* dst = Ctab + ((size_t)blockIdx.y) * Nhp + m;
    if (m >= Nh || a < -w0 || a > w0) { *dst = make_double2(0.0, 0.0); return; }
    const double* arow = sol + (size_t)t * Fab + (size_t)(a + w0) * L1;
    // sum_b arow[b] W1^(m b), b = -w1 .. w1, as  a_0 + sum_{b >= 1} (a_b W^b + a_-b conj(W^b)):  one table entry per column and a
    // running product instead of a 64-bit modulo and a scattered table read per tap (|W| = 1: the product drifts by ~1 ulp a step)
    const cplx w1c = root1[m];                       // m < Nh <= N1
    double cx = arow[w1], cy = 0.0;
    cplx wb = make_double2(1.0, 0.0);
    for (int b = 1; b <= w1; ++b) {
        wb = cmul(wb, w1c);
        const double ap = arow[w1 + b], am = arow[w1 - b];
        cx = fma(ap + am, wb.x, cx);
        cy = fma(ap - am, wb.y, cy);
    }
    if (a == 0) cx -= so;
    *dst = make_double2(f * cx, f * cy);
}

template <int DK, int W, int KS>
__global__ void __launch_bounds__(256, (W > 8 ? 2 : 4)) vconv_mixed(const cplx* __restrict__ stage, cplx* __restrict__ D, const cplx* __restrict__ Ctab,
                                                      const double* __restrict__ kbx, int N0, int Nh, int Nhp, SpecLayout lay,
                                                      cplx* __restrict__ trash)
{
    constexpr int L = 2 * W + 1, NJ = DK + 1, FIJ = (DK + 1) * (DK + 2) / 2, R = KS * L - 2 * W;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];        // FIJ * L * 16 complex (up to 64 KB: dynamic)
    cplx* ctab = reinterpret_cast<cplx*>(smem_raw);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cl = lane & 15, sl = lane >> 4;
    const int m = blockIdx.x * 16 + cl;
    const bool active = m < Nh;
    const int mc = active ? m : Nh - 1;
    for (int e = threadIdx.x; e < FIJ * L * 16; e += 256) {
        const int ta = e >> 4, c = e & 15;
        ctab[e] = Ctab[(size_t)ta * Nhp + (size_t)min((int)blockIdx.x * 16 + c, Nh - 1)];
    }
    __syncthreads();
    const int x0 = ((blockIdx.y * 4 + wv) * 4 + sl) * R;          // first output row of this stream
    if (x0 >= N0) return;                                         // (no barrier below)
    const size_t plane_sz = (size_t)N0 * Nhp, mo = lay.col(mc), rs = (size_t)lay.rstride;
    // acc[q] collects output row (current source row) + q - W; after a source row, acc[0] is complete, is stored, and the
    // window slides down by one (a register rotation: L moves against 24 L FMAs)
    cplx acc[L];
#pragma unroll
    for (int q = 0; q < L; ++q) acc[q] = make_double2(0.0, 0.0);
    int y = x0 - W;
    if (y < 0) y += N0;
#pragma unroll 1
    for (int sI = 0; sI < KS * L; ++sI) {
        // the table does not change along the walk, so the compiler would hoist all FIJ * L reads out of the loop (4 registers
        // each -> scratch spills); an opaque zero in the address makes them re-read from LDS for every source row
        int opq;
        asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
        const cplx* __restrict__ ct = ctab + cl + opq;
        cplx S[NJ];
        double fx[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            S[jj] = stage[(size_t)jj * plane_sz + mo + (size_t)y * rs];
            fx[jj] = kbx[(size_t)jj * N0 + y];
        }
#pragma unroll
        for (int q = 0; q < L; ++q) {           // tap a = q - W
            __builtin_amdgcn_sched_barrier(0);  // ... and a scheduling fence per tap keeps them from all being issued up front
            double ax = acc[q].x, ay = acc[q].y;
#pragma unroll
            for (int jj = 0; jj <= DK; ++jj) {
                double ex = 0.0, ey = 0.0;                         // E_j = sum_i cx^i[y] C'_(i,j)[a]
#pragma unroll
                for (int ii = 0; ii <= DK - jj; ++ii) {
                    const int t = ii * (DK + 1) - (ii * (ii - 1)) / 2 + jj;           // REF_ij order: i major, j = 0 .. DK - i
                    const cplx c = ct[(t * L + q) * 16];
                    ex = fma(fx[ii], c.x, ex);
                    ey = fma(fx[ii], c.y, ey);
                }
                ax = fma(S[jj].x, ex, fma(-S[jj].y, ey, ax));
                ay = fma(S[jj].x, ey, fma(S[jj].y, ex, ay));
            }
            acc[q] = make_double2(ax, ay);
        }
        // the output row completed by this source row.  Stored without a branch (rows before the stream's first output, rows
        // beyond the image and the padding columns go to a per-thread trash slot): a conditional store splits the loop body into
        // blocks, and the table reads then end up a block away from the arithmetic that consumes them, i.e. in scratch
        const int xo = x0 - 2 * W + sI;
        const unsigned long long ok64 = 0ULL - (unsigned long long)(sI >= 2 * W && xo < N0 && active);     // all ones / zero: no select, no branch
        const unsigned long long a_ok = (unsigned long long)(D + mo + (size_t)max(xo, 0) * rs), a_tr = (unsigned long long)(trash + threadIdx.x);
        *reinterpret_cast<cplx*>((a_ok & ok64) | (a_tr & ~ok64)) = acc[0];
#pragma unroll
        for (int q = 0; q + 1 < L; ++q) acc[q] = acc[q + 1];
        acc[L - 1] = make_double2(0.0, 0.0);
        if (++y == N0) y = 0;
    }
}

// vconv_point: the same sum for ONE output element (row x of spectrum column m), taps read straight from the global table -- no sliding
// window.  With Nh = N1 / 2 + 1 the 16-column tiles of vconv_mixed2 end in a tile holding the Nyquist column alone; that column is
// taken point by point instead (a few rows per workgroup of the main launch, or the vconv_direct launch), which leaves the main
// launch a tile count that divides the chip evenly (128 tiles x 4 = two workgroups per CU at 4096^2).
template <int DK, int W>
__device__ __forceinline__ void vconv_point(const cplx* __restrict__ stage, cplx* __restrict__ D, const cplx* __restrict__ Ctab,
                                            const double* __restrict__ kbx, int N0, int Nhp, SpecLayout lay, int x, int m)
{
    constexpr int NJ = DK + 1, L = 2 * W + 1;
    const size_t plane_sz = (size_t)N0 * Nhp, mo = lay.col(m), rs = (size_t)lay.rstride;
    double ax = 0.0, ay = 0.0;
#pragma unroll
    for (int q = 0; q < L; ++q) {               // tap a = q - W: source row y = x - a
        int y = x - (q - W);
        if (y < 0) y += N0;
        if (y >= N0) y -= N0;
        double fx[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) fx[jj] = kbx[(size_t)jj * N0 + y];
#pragma unroll
        for (int jj = 0; jj <= DK; ++jj) {
            const cplx S = stage[(size_t)jj * plane_sz + mo + (size_t)y * rs];
            double ex = 0.0, ey = 0.0;
#pragma unroll
            for (int ii = 0; ii <= DK - jj; ++ii) {
                const int t = ii * (DK + 1) - (ii * (ii - 1)) / 2 + jj;
                const cplx c = Ctab[((size_t)t * L + q) * Nhp + m];
                if (ii == 0) { ex = c.x; ey = c.y; continue; }
                ex = fma(fx[ii], c.x, ex);
                ey = fma(fx[ii], c.y, ey);
            }
            ax = fma(S.x, ex, fma(-S.y, ey, ax));
            ay = fma(S.x, ey, fma(S.y, ex, ay));
        }
    }
    D[mo + (size_t)x * rs] = make_double2(ax, ay);
}
template <int DK, int W>
__global__ void __launch_bounds__(256) vconv_direct(const cplx* __restrict__ stage, cplx* __restrict__ D, const cplx* __restrict__ Ctab,
                                                    const double* __restrict__ kbx, int N0, int Nh, int Nhp, SpecLayout lay, int m0)
{
    const int x = blockIdx.x * 256 + threadIdx.x, m = m0 + blockIdx.y;
    if (x >= N0 || m >= Nh) return;
    vconv_point<DK, W>(stage, D, Ctab, kbx, N0, Nhp, lay, x, m);
}

// vconv_mixed2: the same walk two source rows at a time -- the FIJ table entries of a tap are read from LDS once and serve
// both rows (row y feeds window slot q, row y + 1 slot q + 1), which halves the LDS traffic and doubles the arithmetic behind
// every LDS round trip; the window has L + 1 slots and slides by two.
template <int DK, int W, int KS>
__global__ void __launch_bounds__(256, ((DK <= 2 || W > 8) ? 2 : 3)) vconv_mixed2(const cplx* __restrict__ stage, cplx* __restrict__ D, const cplx* __restrict__ Ctab,
                                                       const double* __restrict__ kbx, int N0, int Nh, int Nhp, SpecLayout lay,
                                                       cplx* __restrict__ trash, int Rrt, int m_direct)
{
    // Rrt > 0: output rows per stream chosen by the host (a whole number of resident rounds, see apply_finish); else KS * L - 2 W
    constexpr int L = 2 * W + 1, NJ = DK + 1, FIJ = (DK + 1) * (DK + 2) / 2;
    constexpr bool PIPE = FIJ <= 6;
    const int R = Rrt > 0 ? Rrt : KS * L - 2 * W, NSRC = (R + 2 * W + 1) / 2 * 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* ctab = reinterpret_cast<cplx*>(smem_raw);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cl = lane & 15, sl = lane >> 4;
    const int m = blockIdx.x * 16 + cl;
    const bool active = m < Nh;
    const int mc = active ? m : Nh - 1;
    for (int e = threadIdx.x; e < FIJ * L * 16; e += 256) {
        const int ta = e >> 4, c = e & 15;
        ctab[e] = Ctab[(size_t)ta * Nhp + (size_t)min((int)blockIdx.x * 16 + c, Nh - 1)];
    }
    __syncthreads();
    if (m_direct < Nh) {
        // this workgroup's share of the columns [m_direct, Nh) that no tile of the launch covers (see vconv_point): a few points per
        // workgroup, done first so that their loads overlap the other resident workgroup's walk
        const int nwg = (int)(gridDim.x * gridDim.y), wg = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int npts = N0 * (Nh - m_direct), per = (npts + nwg - 1) / nwg;
        for (int e = threadIdx.x; e < per; e += 256) {
            const int idx = wg * per + e;
            if (idx < npts) vconv_point<DK, W>(stage, D, Ctab, kbx, N0, Nhp, lay, idx % N0, m_direct + idx / N0);
        }
    }
    const int x0 = ((blockIdx.y * 4 + wv) * 4 + sl) * R;
    if (x0 >= N0) return;
    const size_t plane_sz = (size_t)N0 * Nhp, mo = lay.col(mc), rs = (size_t)lay.rstride;
    cplx acc[L + 1];
#pragma unroll
    for (int q = 0; q <= L; ++q) acc[q] = make_double2(0.0, 0.0);
    int y = x0 - W;
    if (y < 0) y += N0;
    // the two source rows of a step are fetched one step ahead: a wave shares its SIMD with one or two others at most, which
    // does not hide a trip to HBM at the top of every step (measured: 1.1 - 1.6 us of a 6 us step)
    cplx S0[NJ], S1[NJ], T0[NJ], T1[NJ];
    double f0[NJ], f1[NJ], g0[NJ], g1[NJ];
    {
        const int y1 = (y + 1 == N0) ? 0 : y + 1;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            T0[jj] = stage[(size_t)jj * plane_sz + mo + (size_t)y * rs];
            T1[jj] = stage[(size_t)jj * plane_sz + mo + (size_t)y1 * rs];
            g0[jj] = kbx[(size_t)jj * N0 + y];
            g1[jj] = kbx[(size_t)jj * N0 + y1];
        }
        y = (y1 + 1 == N0) ? 0 : y1 + 1;
    }
#pragma unroll 1
    for (int sI = 0; sI < NSRC; sI += 2) {
        int opq;
        asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
        const cplx* __restrict__ ct = ctab + cl + opq;
        const int y1 = (y + 1 == N0) ? 0 : y + 1;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) { S0[jj] = T0[jj]; S1[jj] = T1[jj]; f0[jj] = g0[jj]; f1[jj] = g1[jj]; }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {       // (the rows after the last step's are read and dropped: any row index is valid)
            T0[jj] = stage[(size_t)jj * plane_sz + mo + (size_t)y * rs];
            T1[jj] = stage[(size_t)jj * plane_sz + mo + (size_t)y1 * rs];
            g0[jj] = kbx[(size_t)jj * N0 + y];
            g1[jj] = kbx[(size_t)jj * N0 + y1];
        }
        // PIPE (up to six terms): the table entries of tap q + 1 are read from LDS while tap q is computed; the scheduling fences
        // keep exactly one tap of reads in flight (without them the compiler issues all FIJ * L reads up front and spills)
        cplx cn[PIPE ? FIJ : 1];
        if (PIPE) {
#pragma unroll
            for (int t = 0; t < FIJ; ++t) cn[t] = ct[(t * L + 0) * 16];
        }
#pragma unroll
        for (int q = 0; q < L; ++q) {           // tap a = q - W
            __builtin_amdgcn_sched_barrier(0);
            cplx cc[PIPE ? FIJ : 1];
            if (PIPE) {
#pragma unroll
                for (int t = 0; t < FIJ; ++t) cc[t] = cn[t];
                if (q + 1 < L) {
#pragma unroll
                    for (int t = 0; t < FIJ; ++t) cn[t] = ct[(t * L + q + 1) * 16];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            double ax = acc[q].x, ay = acc[q].y, bx = acc[q + 1].x, by = acc[q + 1].y;
#pragma unroll
            for (int jj = 0; jj <= DK; ++jj) {
                double ex = 0.0, ey = 0.0, gx = 0.0, gy = 0.0;
#pragma unroll
                for (int ii = 0; ii <= DK - jj; ++ii) {
                    const int t = ii * (DK + 1) - (ii * (ii - 1)) / 2 + jj;
                    const cplx c = PIPE ? cc[PIPE ? t : 0] : ct[(t * L + q) * 16];
                    if (ii == 0) { ex = c.x; ey = c.y; gx = c.x; gy = c.y; continue; }        // cx^0 = 1 (polynomial plans only take this path)
                    ex = fma(f0[ii], c.x, ex); ey = fma(f0[ii], c.y, ey);
                    gx = fma(f1[ii], c.x, gx); gy = fma(f1[ii], c.y, gy);
                }
                ax = fma(S0[jj].x, ex, fma(-S0[jj].y, ey, ax));
                ay = fma(S0[jj].x, ey, fma(S0[jj].y, ex, ay));
                bx = fma(S1[jj].x, gx, fma(-S1[jj].y, gy, bx));
                by = fma(S1[jj].x, gy, fma(S1[jj].y, gx, by));
            }
            acc[q] = make_double2(ax, ay);
            acc[q + 1] = make_double2(bx, by);
        }
        // the two output rows completed by these source rows (branch-free stores, see vconv_mixed)
        const int xo = x0 - 2 * W + sI;
        {
            const unsigned long long ok64 = 0ULL - (unsigned long long)(sI >= 2 * W && xo < N0 && xo < x0 + R && active);
            const unsigned long long a_ok = (unsigned long long)(D + mo + (size_t)max(xo, 0) * rs), a_tr = (unsigned long long)(trash + threadIdx.x);
            *reinterpret_cast<cplx*>((a_ok & ok64) | (a_tr & ~ok64)) = acc[0];
        }
        {
            const int x1 = xo + 1;
            const unsigned long long ok64 = 0ULL - (unsigned long long)(sI + 1 >= 2 * W && x1 < N0 && x1 < x0 + R && active);
            const unsigned long long a_ok = (unsigned long long)(D + mo + (size_t)max(x1, 0) * rs), a_tr = (unsigned long long)(trash + threadIdx.x);
            *reinterpret_cast<cplx*>((a_ok & ok64) | (a_tr & ~ok64)) = acc[1];
        }
#pragma unroll
        for (int q = 0; q + 2 <= L; ++q) acc[q] = acc[q + 2];
        acc[L - 1] = make_double2(0.0, 0.0);
        acc[L] = make_double2(0.0, 0.0);
        y = (y1 + 1 == N0) ? 0 : y1 + 1;
    }
}

// vconv_tensor: the mixed-domain apply for a full TENSOR basis (B-spline kernels: term t = ii NJ + jj is row factor ii x column factor jj,
// BSplineSFFT.py's REF_ij order).  The closed form of the inverse column transform holds for any separable term:
//   sum_l FI_t[l][m] W0^(l a) e^(+2 pi i l x / N0) = N0 * bx_ii[x - a] * S_jj[x - a][m],     S_jj = row-DFT(I * by_jj),
// so the apply pass of a B-spline plan needs NJ row transforms instead of NI NJ plane transforms and no column transform at all
// (config 3, 5 x 5 terms at 6144^2: 25 forward plane transforms + the inverse column pass, 12 + 2 ms, become 5 row transforms and this
// kernel).  Same walk as vconv_mixed2 -- two source rows per table read, a sliding window of L + 1 accumulators -- with
// E_jj = sum_ii bx_ii[y] C'_(ii,jj)[a] over all NI row factors (no factor is identically one here).  A wave is CT columns x 64 / CT
// row streams; CT = 8 keeps the NI NJ L CT table of a workgroup at 54 KB for 5 x 5 terms (two workgroups per CU).
// NA < NI: at most NA CONSECUTIVE row factors are nonzero on any row (B-splines of degree NA - 1), the first of them listed per row in
// `ibase`: the sum over ii runs over those only -- 3 x 4 + 8 instead of 5 x 4 + 8 multiply-adds per (tap, column factor, row pair) for
// the 5 x 5 quadratic basis of config 3.  The two rows of a step have a base each (they differ at a knot).
template <int NI, int NJ, int W, int CT, int NA = NI>
__global__ void __launch_bounds__(256, 2) vconv_tensor(const cplx* __restrict__ stage, cplx* __restrict__ D, const cplx* __restrict__ Ctab,
                                                       const double* __restrict__ kbx, const int* __restrict__ ibase, int N0, int Nh, int Nhp,
                                                       SpecLayout lay, cplx* __restrict__ trash, int R)
{
    constexpr bool SPARSE = NA < NI;
    constexpr int L = 2 * W + 1, FIJ = NI * NJ, SPW = 64 / CT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* ctab = reinterpret_cast<cplx*>(smem_raw);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cl = lane % CT, sl = lane / CT;
    const int m = blockIdx.x * CT + cl;
    const bool active = m < Nh;
    const int mc = active ? m : Nh - 1;
    for (int e = threadIdx.x; e < FIJ * L * CT; e += 256) {
        const int ta = e / CT, c = e % CT;
        ctab[e] = Ctab[(size_t)ta * Nhp + (size_t)min((int)blockIdx.x * CT + c, Nh - 1)];
    }
    __syncthreads();
    const int NSRC = (R + 2 * W + 1) / 2 * 2;
    const int x0 = ((blockIdx.y * 4 + wv) * SPW + sl) * R;
    if (x0 >= N0) return;
    const size_t plane_sz = (size_t)N0 * Nhp, mo = lay.col(mc), rs = (size_t)lay.rstride;
    cplx acc[L + 1];
#pragma unroll
    for (int q = 0; q <= L; ++q) acc[q] = make_double2(0.0, 0.0);
    int y = x0 - W;
    if (y < 0) y += N0;
    // the two source rows of a step (and their row factors) are fetched ONE STEP AHEAD, as in vconv_mixed2: at two waves per SIMD a trip
    // to HBM at the top of every step is not hidden (round 4: 1.9 ms at config 3 with the loads at the top of the step)
    cplx S0[NJ], S1[NJ], T0[NJ], T1[NJ];
    double f0[NA], f1[NA], g0[NA], g1[NA];
    int nb0 = 0, nb1 = 0;                       // first row factor of the prefetched rows
    {
        const int y1 = (y + 1 == N0) ? 0 : y + 1;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            T0[jj] = stage[(size_t)jj * plane_sz + mo + (size_t)y * rs];
            T1[jj] = stage[(size_t)jj * plane_sz + mo + (size_t)y1 * rs];
        }
        if (SPARSE) { nb0 = ibase[y]; nb1 = ibase[y1]; }
#pragma unroll
        for (int ii = 0; ii < NA; ++ii) {
            g0[ii] = kbx[(size_t)(nb0 + ii) * N0 + y];
            g1[ii] = kbx[(size_t)(nb1 + ii) * N0 + y1];
        }
        y = (y1 + 1 == N0) ? 0 : y1 + 1;
    }
#pragma unroll 1
    for (int sI = 0; sI < NSRC; sI += 2) {
        int opq;
        asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
        const cplx* __restrict__ ct = ctab + cl + opq + nb0 * (NJ * L * CT);
        const cplx* __restrict__ ct1 = ctab + cl + opq + nb1 * (NJ * L * CT);
        const int y1 = (y + 1 == N0) ? 0 : y + 1;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) { S0[jj] = T0[jj]; S1[jj] = T1[jj]; }
#pragma unroll
        for (int ii = 0; ii < NA; ++ii) { f0[ii] = g0[ii]; f1[ii] = g1[ii]; }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {       // (the rows after the last step's are read and dropped: any row index is valid)
            T0[jj] = stage[(size_t)jj * plane_sz + mo + (size_t)y * rs];
            T1[jj] = stage[(size_t)jj * plane_sz + mo + (size_t)y1 * rs];
        }
        if (SPARSE) { nb0 = ibase[y]; nb1 = ibase[y1]; }
#pragma unroll
        for (int ii = 0; ii < NA; ++ii) {
            g0[ii] = kbx[(size_t)(nb0 + ii) * N0 + y];
            g1[ii] = kbx[(size_t)(nb1 + ii) * N0 + y1];
        }
#pragma unroll
        for (int q = 0; q < L; ++q) {           // tap a = q - W
            __builtin_amdgcn_sched_barrier(0);
            double ax = acc[q].x, ay = acc[q].y, bx = acc[q + 1].x, by = acc[q + 1].y;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                double ex = 0.0, ey = 0.0, gx = 0.0, gy = 0.0;
#pragma unroll
                for (int ii = 0; ii < NA; ++ii) {
                    const cplx c = ct[((ii * NJ + jj) * L + q) * CT];
                    const cplx d = SPARSE ? ct1[((ii * NJ + jj) * L + q) * CT] : c;
                    ex = fma(f0[ii], c.x, ex); ey = fma(f0[ii], c.y, ey);
                    gx = fma(f1[ii], d.x, gx); gy = fma(f1[ii], d.y, gy);
                }
                ax = fma(S0[jj].x, ex, fma(-S0[jj].y, ey, ax));
                ay = fma(S0[jj].x, ey, fma(S0[jj].y, ex, ay));
                bx = fma(S1[jj].x, gx, fma(-S1[jj].y, gy, bx));
                by = fma(S1[jj].x, gy, fma(S1[jj].y, gx, by));
            }
            acc[q] = make_double2(ax, ay);
            acc[q + 1] = make_double2(bx, by);
        }
        const int xo = x0 - 2 * W + sI;
        {
            const unsigned long long ok64 = 0ULL - (unsigned long long)(sI >= 2 * W && xo < N0 && xo < x0 + R && active);
            const unsigned long long a_ok = (unsigned long long)(D + mo + (size_t)max(xo, 0) * rs), a_tr = (unsigned long long)(trash + threadIdx.x);
            *reinterpret_cast<cplx*>((a_ok & ok64) | (a_tr & ~ok64)) = acc[0];
        }
        {
            const int x1 = xo + 1;
            const unsigned long long ok64 = 0ULL - (unsigned long long)(sI + 1 >= 2 * W && x1 < N0 && x1 < x0 + R && active);
            const unsigned long long a_ok = (unsigned long long)(D + mo + (size_t)max(x1, 0) * rs), a_tr = (unsigned long long)(trash + threadIdx.x);
            *reinterpret_cast<cplx*>((a_ok & ok64) | (a_tr & ~ok64)) = acc[1];
        }
#pragma unroll
        for (int q = 0; q + 2 <= L; ++q) acc[q] = acc[q + 2];
        acc[L - 1] = make_double2(0.0, 0.0);
        acc[L] = make_double2(0.0, 0.0);
        y = (y1 + 1 == N0) ? 0 : y1 + 1;
    }
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


// ------------------------------------------------------------------------------------------------
// Small spectrum-arithmetic kernels behind the FFT utilities (noise decorrelation, FFT convolution:
// sfft/utils/PureCupyFFTKits.py, PureCupyDeCorrelationCalculator.py)
// ------------------------------------------------------------------------------------------------
// dst(l, m) = f * src(l, m), each side in its own layout (a dense caller array is a row-major layout of leading dimension Nh)
__global__ void copy_spectrum_scaled(const cplx* __restrict__ src, cplx* __restrict__ dst, int N0, int Nh, SpecLayout sl, SpecLayout dl, double f)
{
    const int m = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (m < Nh) { const cplx v = src[sl.at(l, m)]; dst[dl.at(l, m)] = make_double2(v.x * f, v.y * f); }
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
__global__ void copy_spectrum(const cplx* __restrict__ src, cplx* __restrict__ dst, int N0, int Nh, SpecLayout sl)
{
    const int m = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (m < Nh) dst[(size_t)l * Nh + m] = src[sl.at(l, m)];
}

// ------------------------------------------------------------------------------------------------
// Grid-wise space-varying convolution (BSpline_GridConvolve.GSVC_GPU, sfft/BSplineSFFT.py:4951-5006):
//   out(x, y) = sum_ab K[label(x, y)][a][b] * in(x + w0 - a, y + w1 - b),   w = (L - 1) / 2,   zeros beyond the image
// (= scipy / cupyx convolve2d(mode='same', boundary='fill') of every box segment with its own kernel; the reference
// extends each box by w + 1 pixels before convolving, so inside a box only the image border ever supplies zeros).
// 16 x 16 output pixels per workgroup.  The kernel stamp is walked in bands of A rows: the (16 + A - 1) x (16 + L1 - 1) input
// pixels a band needs sit in LDS (A is chosen by the host so that a band fits 64 KB: any stamp size runs, e.g. the 411 x 411
// decorrelation kernels of the reference's NIRCam example, where the whole halo would be 1.4 MB).  A wave reads the stamp of
// its label through scalar loads; a wave that straddles box borders walks the band once per distinct label among its lanes
// with the other lanes' sums discarded, so the stamp reads stay scalar.
// Two partial sums per pixel (even / odd stamp columns) keep two fp64 FMA chains in flight.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) grid_convolve(const double* __restrict__ in, const int* __restrict__ labels,
                                                     const double* __restrict__ kers, int N0, int N1, int Nseg, int L0, int L1,
                                                     int A, double* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* tile = reinterpret_cast<double*>(smem_raw);
    const int w0 = (L0 - 1) / 2, w1 = (L1 - 1) / 2;
    const int TW = 16 + L1 - 1;
    const int x0 = blockIdx.y * 16, y0 = blockIdx.x * 16;
    const int oy = y0 - (L1 - 1 - w1);                                  // image column of tile[.][0]
    const int lx = threadIdx.x >> 4, ly = threadIdx.x & 15;
    const int x = x0 + lx, y = y0 + ly;
    const bool inside = x < N0 && y < N1;
    int lab = inside ? labels[(size_t)x * N1 + y] : -1;
    if (lab >= Nseg) lab = -1;
    double acc = 0.0;
    for (int a0 = 0; a0 < L0; a0 += A) {
        const int na = min(A, L0 - a0);
        // rows of the image this band touches: x + w0 - a for a in [a0, a0 + na) and the 16 rows of the block
        const int TH = 16 + na - 1;
        const int ox = x0 + w0 - (a0 + na - 1);                         // image row of tile[0][.]
        __syncthreads();
        for (int e = threadIdx.x; e < TH * TW; e += 256) {
            const int tx = e / TW, ty = e - tx * TW;
            const int gx = ox + tx, gy = oy + ty;
            tile[e] = (gx >= 0 && gx < N0 && gy >= 0 && gy < N1) ? in[(size_t)gx * N1 + gy] : 0.0;
        }
        __syncthreads();
        int pending = lab;
        for (;;) {                                                      // one walk of the band per distinct label of this wave
            const unsigned long long m = __ballot(pending >= 0);
            if (!m) break;
            const int cur = __builtin_amdgcn_readlane(pending, __ffsll((long long)m) - 1);      // wave-uniform
            const double* __restrict__ K = kers + ((size_t)cur * L0 + a0) * L1;
            double s0 = 0.0, s1 = 0.0;
            for (int da = 0; da < na; ++da) {
                // tile row of in(x + w0 - (a0 + da), .) is lx + (na - 1) - da; column of in(., y + w1 - b) is ly + (L1 - 1) - b
                const double* trow = tile + (lx + (na - 1) - da) * TW + ly + (L1 - 1);
                const double* __restrict__ Kr = K + (size_t)da * L1;
                int b = 0;
                for (; b + 1 < L1; b += 2) {
                    s0 = fma(Kr[b], trow[-b], s0);
                    s1 = fma(Kr[b + 1], trow[-b - 1], s1);
                }
                if (b < L1) s0 = fma(Kr[b], trow[-b], s0);
            }
            if (pending == cur) { acc += s0 + s1; pending = -1; }
        }
    }
    if (inside) out[(size_t)x * N1 + y] = acc;
}

#endif
