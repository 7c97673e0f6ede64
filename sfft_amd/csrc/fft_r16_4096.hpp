// fft_r16_4096.hpp -- register-resident radix-16 fast path for 4096-point axes.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_FFT_R16_4096_HPP
#define SFFT_AMD_FFT_R16_4096_HPP

// ================================================================================================
// Fast path for 4096-point axes: register-resident radix-16 FFT.  256 threads own 16 points each through
// three radix-16 stages (4096 = 16^3); LDS is used only for the two inter-stage exchanges (padded by one
// element per 16 so that the stride-16 writes of stage 1 are conflict free), not as the working array.
// ================================================================================================
#define F4K_LDS 4352                                 // 4096 + 4096/16 complex per transform


// 4096-point forward FFT.  In: u[r] = x[j + 256 r].  Out: u[R16_OUT(s)] = X[j + 256 s].  j in [0, 256).
// `lds` = this transform's F4K_LDS-element scratch.  Every thread of the block must call (barriers inside).
// sw (0 or 4): element x of the exchanges lives at slot pad16(x ^ sw).  Kernels that run TWO transforms side by side in neighbouring
// lanes (lane = 2 j + c, the column-pair kernels) give the second one sw = 4 and a region a whole number of 256-byte bank rows away:
// a 16-byte LDS store is served in groups of 8 contiguous lanes against 128-byte bank rows, a 16-byte load in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) against 256-byte rows (MI355X_MICROARCH.md, LDS).  With j in the lane's upper
// bits a group holds the slots {0,1,6,7,10,11,12,13} (or their complement) of both transforms: the regions' old 64-byte skew kept the
// stores apart and left every load 2-way conflicted (SQ_LDS_BANK_CONFLICT = 2.0 x SQ_INSTS_LDS, profiles/r04_h_pmc_stall_cfg2.txt);
// flipping bit 2 of the slot index maps that set onto its complement AND moves the stores by 64 bytes: no conflict either way.
__device__ __forceinline__ void fft4096_core(cplx (&u)[16], int j, cplx* lds, const cplx* __restrict__ tw, int sw = 0)
{
    dft16(u);
    {
        cplx* wA = lds + 17 * j + sw;                // slots sx ^ sw: + sw where bit 2 of sx is clear, - sw where it is set
        cplx* wB = lds + 17 * j - sw;
#pragma unroll
        for (int sx = 0; sx < 16; ++sx) ((sx & 4) ? wB : wA)[sx] = u[R16_OUT(sx)];          // pad16((16 j + sx) ^ sw)
    }
    __syncthreads();
    const cplx* rd = lds + pad16(j ^ sw);            // pad16((j + 256 r) ^ sw) = pad16(j ^ sw) + 272 r
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = rd[272 * r];
    __syncthreads();
    const int k = j & 15;
    twiddle16(u, tw, 16 * k);
    dft16(u);
    {
        cplx* w2 = lds + pad16((j - k) * 16 + (k ^ sw));        // pad16(((j - k) 16 + k + 16 sx) ^ sw) = ... + 17 sx
#pragma unroll
        for (int sx = 0; sx < 16; ++sx) w2[17 * sx] = u[R16_OUT(sx)];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = rd[272 * r];
    twiddle16(u, tw, j);
    dft16(u);
}

// 2 x 2 exchange between the lane pairs of a quad (DPP quad_perm): see cols_fwd_weighted_4096_q and the pair-major stores of rows_r2c_4096
__device__ __forceinline__ double dpp_swap2(double x)          // the value held by lane ^ 2
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);     // quad_perm [2, 3, 0, 1]
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ cplx dpp_swap2(cplx v) { return make_double2(dpp_swap2(v.x), dpp_swap2(v.y)); }
__device__ __forceinline__ cplx csel(bool p, cplx a, cplx b) { return make_double2(p ? a.x : b.x, p ? a.y : b.y); }

// rows, real -> half complex (N1 = 4096), two image rows per transform, spatial factors fused.  Planes [first, first +
// count) of a launch group share their source image: the workgroup reads its two rows once and produces every plane.
#define ROWMOM_FUSED_MAX 8                           // most moments per row the row pass computes itself (polynomial bases: DK + DB + 1 <= 7)
struct RowGroups {
    int ngroups; int first[SFFT_MAX_PLANES]; int count[SFFT_MAX_PLANES];
    // row moments of a group's source image, sum_n src[l][n] * cy(n)^q for q < mom_nq (cy = (n + 1) / N1), written to mom_out[l][SFFT_MAX_BQ]:
    // what row_moments computes in a pass of its own over the same image (Delta and the real-space Gamma block), here from the
    // rows the workgroup has just loaded.  mom_nq = 0: none.
    double* mom_out[SFFT_MAX_PLANES]; int mom_nq[SFFT_MAX_PLANES];
};

// Workgroup b takes row pair (b % 8) * pairs_per_xcd + b / 8: consecutive row pairs run on one XCD, so that with a panel
// layout the pieces of a 128-byte line written by neighbouring row pairs merge in that XCD's L2.
// pm (pair-major lines, the stage planes of cols_fwd_weighted_4096_z): the 128-byte line of rows 2p, 2p + 1 of a 4-column panel is stored as
// [column pair h][row parity][column c] instead of [row parity][4 columns]; this kernel writes whole lines either way.
__global__ void __launch_bounds__(256, 2) rows_r2c_4096(RowsArgs a, RowGroups grp, cplx* __restrict__ out, int N0, int Nhp, SpecLayout lay,
                                                     const cplx* __restrict__ tw, double scale, int pairs_per_xcd, int pm)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N1 = 4096;
    const int j = threadIdx.x;
    const int pfirst = grp.first[blockIdx.y], pcount = grp.count[blockIdx.y];
    const int rp = (int)(blockIdx.x & 7) * pairs_per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= pairs_per_xcd || 2 * rp >= N0) return;
    const int l0 = 2 * rp, l1 = l0 + 1;
    const bool has1 = l1 < N0;
    const int m0 = SFFT_CR(l0, 32), m1 = SFFT_CR(l1, 32);            // (the rows whose memory is touched: l0, l1 in the product)
    const double* __restrict__ src = a.src[pfirst];
    const double* r0p = src + (size_t)m0 * N1;
    const double* r1p = src + (size_t)(has1 ? m1 : m0) * N1;
    // (selects on wave-uniform conditions inside these unrolled loops become one scalar branch per element: the second row
    //  is read unconditionally -- r1p falls back to row l0 -- and scaled by 0, and missing weights point at a table of ones)
    const double h1 = has1 ? 1.0 : 0.0;
    double x0[16], x1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = j + 256 * r;
        x0[r] = ld_stream(r0p + n);
        x1[r] = ld_stream(r1p + n) * h1;
    }
    const int mnq = grp.mom_nq[blockIdx.y];
    if (mnq > 0) {                          // (workgroup uniform)
        double a0[ROWMOM_FUSED_MAX], a1[ROWMOM_FUSED_MAX];
#pragma unroll
        for (int q = 0; q < ROWMOM_FUSED_MAX; ++q) a0[q] = a1[q] = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double c = (double)(j + 256 * r + 1) * (1.0 / 4096.0);
            double pw = 1.0;
#pragma unroll
            for (int q = 0; q < ROWMOM_FUSED_MAX; ++q) { a0[q] = fma(x0[r], pw, a0[q]); a1[q] = fma(x1[r], pw, a1[q]); pw *= c; }
        }
        double* red = reinterpret_cast<double*>(lds);         // [4 waves][2 rows][SFFT_MAX_BQ]
#pragma unroll
        for (int q = 0; q < ROWMOM_FUSED_MAX; ++q) {
            if (q < mnq) {
                double u0 = a0[q], u1 = a1[q];
                for (int off = 32; off > 0; off >>= 1) { u0 += __shfl_down(u0, off); u1 += __shfl_down(u1, off); }
                if ((j & 63) == 0) { red[((j >> 6) * 2 + 0) * SFFT_MAX_BQ + q] = u0; red[((j >> 6) * 2 + 1) * SFFT_MAX_BQ + q] = u1; }
            }
        }
        __syncthreads();
        if (j < 2 * mnq) {
            const int row = j / mnq, q = j - row * mnq;
            const double v = red[(0 * 2 + row) * SFFT_MAX_BQ + q] + red[(1 * 2 + row) * SFFT_MAX_BQ + q]
                           + red[(2 * 2 + row) * SFFT_MAX_BQ + q] + red[(3 * 2 + row) * SFFT_MAX_BQ + q];
            if (row == 0 || has1) grp.mom_out[blockIdx.y][(size_t)(l0 + row) * SFFT_MAX_BQ + q] = v;
        }
        __syncthreads();                    // the transform below reuses this LDS
    }
    const double hs = 0.5 * scale;
    for (int pp = 0; pp < pcount; ++pp) {
        const int plane = pfirst + pp;
        const double* __restrict__ wx = a.wx[plane];
        const double* __restrict__ wy = a.wy[plane];
        const double cx0 = wx[l0];
        const double cx1 = wx[has1 ? l1 : l0];
        cplx u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double cyp = wy[j + 256 * r];
            u[r] = make_double2(x0[r] * (cx0 * cyp), x1[r] * (cx1 * cyp));
        }
        if (pp > 0) __syncthreads();            // the previous plane's partner reads are done
        // (an offset the compiler cannot see through: otherwise the 30 stage twiddles are hoisted out of the plane loop,
        //  which costs 120 registers and halves the occupancy)
        int zoff;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
        fft4096_core(u, j, lds, tw + zoff);
        __syncthreads();
#pragma unroll
        for (int sx = 0; sx < 16; ++sx) lds[j + 256 * sx] = u[R16_OUT(sx)];
        __syncthreads();
        cplx* o0 = out + (size_t)plane * N0 * Nhp + (pm ? (size_t)(m0 >> 1) * 8 : (size_t)m0 * lay.rstride);
        cplx* o1 = out + (size_t)plane * N0 * Nhp + (size_t)m1 * lay.rstride;
        if (pm) {       // (launch uniform) pair-major lines: a lane quad stores the 64-byte sector [row l0: c, c + 1][row l1: c, c + 1] of column pair 0, then of pair 1
            const bool even = (j & 2) == 0;
            cplx* ob = o0 + (j & 3);
#pragma unroll
            for (int sx = 0; sx <= 8; ++sx) {
                const int m = j + 256 * sx;
                if (sx < 8 || j < 4) {              // (sx = 8: the quad of column N1 / 2; its other three columns are the panel's padding)
                    const cplx z = u[R16_OUT(sx)];
                    const cplx zp = lds[(N1 - m) & (N1 - 1)];
                    const cplx zc = make_double2(zp.x, -zp.y);
                    const cplx X0 = make_double2(hs * (z.x + zc.x), hs * (z.y + zc.y)), X1 = make_double2(hs * (z.y - zc.y), -hs * (z.x - zc.x));
                    const cplx got = dpp_swap2(csel(even, X1, X0));
                    cplx* oq = ob + (size_t)(m >> 2) * (size_t)lay.pstride;
                    st_stream(oq, csel(even, X0, got));
                    st_stream(oq + 4, csel(even, got, X1));
                }
            }
            continue;
        }
#pragma unroll
        for (int sx = 0; sx <= 8; ++sx) {
            const int m = j + 256 * sx;
            if (sx < 8 || j == 0) {
                const cplx z = u[R16_OUT(sx)];
                const cplx zp = lds[(N1 - m) & (N1 - 1)];
                const cplx zc = make_double2(zp.x, -zp.y);
                const size_t mo = lay.col(m);
                st_stream(o0 + mo, make_double2(hs * (z.x + zc.x), hs * (z.y + zc.y)));
                if (has1) st_stream(o1 + mo, make_double2(hs * (z.y - zc.y), -hs * (z.x - zc.x)));
            }
        }
    }
}

// columns, complex -> complex in place (N0 = 4096), two adjacent columns per workgroup (512 threads).
// Blocks that share 128-byte lines are mapped to the same XCD (block b runs on XCD b % 8) so its L2 merges them.
__global__ void __launch_bounds__(512) cols_c2c_4096(cplx* __restrict__ data, int ncols, int Nhp, SpecLayout lay, const cplx* __restrict__ tw,
                                                     int inverse, double scale, int pairs_per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N0 = 4096;
    const int c = threadIdx.x & 1, j = threadIdx.x >> 1;
    const int cp = (blockIdx.x & 7) * pairs_per_xcd + (blockIdx.x >> 3);
    const int col = 2 * cp + c;
    const bool ok = (blockIdx.x >> 3) < pairs_per_xcd && col < ncols;
    cplx* __restrict__ base = data + (size_t)blockIdx.y * N0 * Nhp + lay.col(ok ? col : 0);
    const size_t rs = (size_t)lay.rstride;
    cplx u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        cplx z = ok ? base[(size_t)(j + 256 * r) * rs] : make_double2(0.0, 0.0);
        if (inverse) z.y = -z.y;
        u[r] = z;
    }
    fft4096_core(u, j, lds + c * F4K_LDS, tw, 4 * c);     // (second column: slots flipped in bit 2, see fft4096_core)
    if (!ok) return;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) {
        cplx z = u[R16_OUT(sx)];
        if (inverse) z.y = -z.y;
        base[(size_t)(j + 256 * sx) * rs] = make_double2(z.x * scale, z.y * scale);
    }
}

// Forward column pass of the weighted planes.  Every spatial term is  I * fx(row) * fy(col):  the row pass only applies fy
// (one "stage" plane per distinct column factor), and here each output plane (fx, fy) is the column transform of its stage
// plane times fx(row).  For the polynomial basis of order 2 that is 3 row transforms for 6 output planes (4 for 7 with J).
// One workgroup per (tile, output).  The 1-D grid is de-interleaved per XCD with the output index fastest, so the
// workgroups that read the same stage tile -- and the neighbouring tiles that share its 128-byte lines -- run back to back
// on one XCD: the tile comes from HBM once and from that XCD's L2 afterwards.

__global__ void __launch_bounds__(512) cols_fwd_weighted_4096(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int ncols,
                                                              int Nhp, SpecLayout lay, const cplx* __restrict__ tw, int npairs)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N0 = 4096;
    const int c = threadIdx.x & 1, j = threadIdx.x >> 1;
    const int total = npairs * g.nout;
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || logical >= total) return;
    const int cp = logical / g.nout, o = logical - cp * g.nout;
    const int col = 2 * cp + c;
    const bool ok = col < ncols;
    const size_t plane_sz = (size_t)N0 * Nhp, cofs = lay.col(ok ? col : 0), rs = (size_t)lay.rstride;
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + cofs;
    const double* __restrict__ w = g.wx[o];
    cplx u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const cplx z = src[(size_t)(j + 256 * r) * rs];         // (!ok: column 0 again; nothing is stored)
        const double f = w[j + 256 * r];
        u[r] = make_double2(z.x * f, z.y * f);
    }
    fft4096_core(u, j, lds + c * F4K_LDS, tw, 4 * c);
    if (!ok) return;
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * plane_sz + cofs;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) dst[(size_t)(j + 256 * sx) * rs] = u[R16_OUT(sx)];
}

// The same pass with FOUR columns (one 64-byte row piece of a 4-column panel) per workgroup.  With two columns per workgroup
// every load / store instruction touches 32 half-used 64-byte sectors and the L1 is the limit (PMC: TCP_PENDING_STALL 55 % of
// the kernel, 29 M 32-byte write requests per launch).  Here a lane quad moves one whole 64-byte piece per instruction -- rows
// 2p and 2p + 1 in two instructions -- and a 2 x 2 exchange between lane pairs (DPP quad_perm) turns that into the
// (row, column pair) ownership of the transform: lane pair (4p, 4p + 1) owns row 2p, pair (4p + 2, 4p + 3) row 2p + 1, first of
// columns (0, 1), then of columns (2, 3).  The two column pairs are transformed one after the other through the same LDS.
__global__ void __launch_bounds__(512) cols_fwd_weighted_4096_q(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int Nhp,
                                                                SpecLayout lay, const cplx* __restrict__ tw, int nquads)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N0 = 4096;
    const int tid = threadIdx.x, c = tid & 1, j = tid >> 1, q4 = tid & 3;
    const bool even = q4 < 2;                                   // this lane's row j is the even one of its quad
    const int total = nquads * g.nout;
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || logical >= total) return;
    const int cq = logical / g.nout, o = logical - cq * g.nout;
    // (the panel's padding columns beyond Nh exist in memory: they are transformed and stored like the others, nobody reads them)
    const size_t plane_sz = (size_t)N0 * Nhp, cofs = (size_t)SFFT_CR(cq, 8) * (size_t)lay.pstride + q4;
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + cofs;
    const double* __restrict__ w = g.wx[o];
    const int rowA = 2 * (j >> 1);
    cplx u1[16], u2[16];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {        // two batches of 8 rows (16 + 8 loads in flight): bounds the registers of this phase
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
    fft4096_core(u1, j, mylds, tw, 4 * c);
    __syncthreads();                                            // the last exchange of the first pair has been read
    fft4096_core(u2, j, mylds, tw, 4 * c);
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * plane_sz + cofs;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) {
        const cplx o1 = u1[R16_OUT(sx)], o2 = u2[R16_OUT(sx)];
        const cplx got = dpp_swap2(csel(even, o2, o1));
        st_stream(dst + (size_t)(rowA + 256 * sx) * 4, csel(even, o1, got));
        st_stream(dst + (size_t)(rowA + 1 + 256 * sx) * 4, csel(even, got, o2));
    }
}

// ------------------------------------------------------------------------------------------------
// Forward column pass, round 6: TWO workgroups per CU.  Where cols_fwd_weighted_4096_q's time goes (scripts/micro/cols_q_ablate.hip, one launch of
// config 2): the kernel 0.40 ms; a pure mover with its loads and stores 0.27 - 0.30; its transforms alone (tile from L2, nothing stored) 0.26 -- one
// 512-thread workgroup holds a CU's whole register file (4 columns x 4096 points = 256 KB) and both 69.6 KB exchange buffers, so load, transform
// and store of a tile run one after the other and nothing overlaps them.  Here a 512-thread workgroup (lane = 2 j + c, 16 points per thread,
// <= 128 registers) transforms ONE column pair; real and imaginary parts go through the LDS one after the other (2 x 34 KB per workgroup), so two
// workgroups -- 16 waves, two tiles in different phases -- fit a CU.  What makes a column pair affordable is the layout on both sides:
//   in   the stage planes keep their 4-column panels, with every 128-byte line (rows 2p, 2p + 1 x columns 0..3) stored pair-major,
//        [pair h][row parity][column c]: a lane quad reads one whole 64-byte sector; the sibling pair's workgroup (next in the grid, same XCD)
//        takes the other half of the line out of the L2.  The row pass writes whole lines either way (rows_r2c_4096, pm = 1).
//   out  the spectra go out in 2-column panels [Nhp / 2][N0][2]: a wave stores 1 KB contiguous.  The Omega / Theta launches read a 16-column
//        tile of 4 rows per step -- 8 full lines in either panel width (measured + 5 % for them, - 20 % for this pass).
// Measured in the micro-benchmark: 0.305 - 0.34 ms against 0.40 (transforms alone 0.16 against 0.26).
// Region c of the LDS starts at c * Z4K_LDS doubles and holds element x at pad16(x ^ 8 c): every 8-byte access is conflict free
// (scripts/lds_conflicts.py).
// ------------------------------------------------------------------------------------------------
#define Z4K_LDS 4368
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
    for (int r = 0; r < 16; ++r) u[r].x = rd[272 * r];              // (the real parts are dead once written: overwritten in place)
    __syncthreads();
    // the imaginary parts still sit in dft16's output order, u[R16_OUT(sx)].y, untouched by the reads above
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

// pstride4: elements between the 4-column panels of a stage plane.  Workgroup order: output fastest, then the two pairs of a panel --
// the 2 * nout workgroups that touch one stage panel run back to back on one XCD.
__global__ void __launch_bounds__(512, 4) cols_fwd_weighted_4096_z(const cplx* __restrict__ stage, cplx* __restrict__ out, ColOuts g, int Nhp,
                                                                   long long pstride4, const cplx* __restrict__ tw, int npairs)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int N0 = 4096;
    const int tid = threadIdx.x, c = tid & 1, j = tid >> 1;
    const int total = npairs * g.nout;
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || logical >= total) return;
    const int cp = logical / g.nout, o = logical - cp * g.nout;
    const int cpm = SFFT_CR(cp, 16);                 // (the pair whose memory is touched: cp in the product)
    const size_t plane_sz = (size_t)N0 * Nhp;
    // element (row l, pair h of panel P, column c) of a stage plane: P * pstride4 + (l >> 1) * 8 + h * 4 + (l & 1) * 2 + c;  row j + 256 r of
    // lane 2 j + c: (tid >> 2) * 8 + (tid & 3) + 1024 r
    const cplx* __restrict__ src = stage + (size_t)g.stage_plane[o] * plane_sz + (size_t)(cpm >> 1) * (size_t)pstride4 + (size_t)((cpm & 1) * 4)
                                   + (size_t)((tid >> 2) * 8 + (tid & 3));
    const double* __restrict__ w = g.wx[o];
    cplx u[16];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {            // two batches of 8 rows: bounds the registers of the load phase
        double f[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { u[8 * hb + r] = ld_stream(src + (size_t)(1024 * (8 * hb + r))); f[r] = w[j + 256 * (8 * hb + r)]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) { u[8 * hb + r].x *= f[r]; u[8 * hb + r].y *= f[r]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    fft4096_core_split2(u, j, lds + c * Z4K_LDS, tw, 8 * c);
    // 2-column panels: element (row l, column c of pair cp) at cp * N0 * 2 + l * 2 + c;  row j + 256 sx of lane 2 j + c: tid + 512 sx
    cplx* __restrict__ dst = out + (size_t)g.out_plane[o] * plane_sz + (size_t)cpm * (size_t)(N0 * 2) + tid;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) st_nt(dst + 512 * sx, u[R16_OUT(sx)]);
}

// rows, half complex -> real (N1 = 4096), two rows per transform, DIFF epilogue (see rows_c2r_diff)
template <int NQ>
__global__ void __launch_bounds__(256) rows_c2r_diff_4096(const cplx* __restrict__ FD, const double* __restrict__ J,
                                                          const double* __restrict__ bpq, BkgArgs bk, double* __restrict__ DIFF,
                                                          int N0, SpecLayout lay, const cplx* __restrict__ tw)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N1 = 4096;
    const int j = threadIdx.x;
    const int l0 = 2 * blockIdx.x, l1 = l0 + 1;
    const bool has1 = l1 < N0;
    const int m0 = SFFT_CR(l0, 32), m1 = SFFT_CR(l1, 32);            // (the rows whose memory is touched: l0, l1 in the product)
    const cplx* f0 = FD + (size_t)m0 * lay.rstride;
    const cplx* f1 = FD + (size_t)(has1 ? m1 : m0) * lay.rstride;
    cplx u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = j + 256 * r;
        const bool mir = m > N1 / 2;
        const int mm = mir ? N1 - m : m;
        const size_t mo = lay.col(mm);
        cplx x0 = ld_stream(f0 + mo);
        cplx x1 = has1 ? ld_stream(f1 + mo) : make_double2(0.0, 0.0);
        if (mm == 0 || mm == N1 / 2) { x0.y = 0.0; x1.y = 0.0; }
        if (mir) { x0.y = -x0.y; x1.y = -x1.y; }
        u[r] = make_double2(x0.x - x1.y, -(x0.y + x1.x));      // conj(X0 + i X1)
    }
    fft4096_core(u, j, lds, tw);
    double c0[NQ], c1[NQ];
    bkg_row_coeffs<NQ>(bk, bpq, l0, N0, c0);
    bkg_row_coeffs<NQ>(bk, bpq, has1 ? l1 : l0, N0, c1);
    const double* j0 = J + (size_t)m0 * N1;
    const double* j1 = J + (size_t)(has1 ? m1 : m0) * N1;
    double* d0 = DIFF + (size_t)m0 * N1;
    double* d1 = DIFF + (size_t)(has1 ? m1 : m0) * N1;
    // The epilogue in batches: all J and background-table loads of a batch first, then its arithmetic and stores.  Written
    // element by element, every table load waits for the previous DIFF store (the table pointer may alias DIFF for all the
    // compiler knows): 32 dependent round trips per thread, 136 us for the kernel instead of 85.
    constexpr int BS = (NQ <= 4) ? 4 : 1;
#pragma unroll
    for (int b0 = 0; b0 < 16; b0 += BS) {
        double jv0[BS], jv1[BS], tb[BS][NQ];
#pragma unroll
        for (int e = 0; e < BS; ++e) {
            const int n = j + 256 * (b0 + e);
            jv0[e] = ld_stream(j0 + n);
            jv1[e] = ld_stream(j1 + n);
#pragma unroll
            for (int q = 0; q < NQ; ++q) tb[e][q] = bk.tby[(size_t)min(q, bk.nq - 1) * N1 + n];      // clamped: always valid
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < BS; ++e) {
            const int n = j + 256 * (b0 + e);
            const cplx z = u[R16_OUT(b0 + e)];
            double B0 = 0.0, B1 = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const bool on = q < bk.nq;
                B0 = fma(on ? c0[q] : 0.0, tb[e][q], B0);
                B1 = fma(on ? c1[q] : 0.0, tb[e][q], B1);
            }
            st_nt(d0 + n, jv0[e] - B0 - z.x);
            if (has1) st_nt(d1 + n, jv1[e] - B1 + z.y);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

#endif
