// lu.hpp -- dense solve with the reference's own semantics: LU with partial (row) pivoting, blocked for the matrix cores.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only (after solver.hpp).
//
// The reference solves every system by pivoted LU (np.linalg.solve / cupy.linalg.solve -> getrf + getrs,
// sfft/sfftcore/SFFTSubtract.py:15-23, 398-403, 743-747).  Here LU takes the systems whose Cholesky attempt meets a non-positive
// pivot, and every system when sfft_set_force_lu(1) is set.  Rounds 1 - 4 ran it unblocked (one pivot launch + one rank-1 launch per
// column: 3 470 launches at n = 1735).  Now:
//
//   A is [(n + 1)][ld] row-major, rows < n, columns <= n; column n is the right-hand side and rides along as one more column of
//   the trailing matrix, so L^-1 P b is a by-product and L is never needed again.  Right-looking over panels of LU_NB = 64 columns:
//
//   lu_panel<W, R>   ONE workgroup factors the (n - k0) x 64 panel with true partial pivoting (the pivot of column j is the largest
//                    |entry| of the UPDATED column j over all rows below, ties to the first row -- idamax's rule).  A pivot search is
//                    a chain of n dependent global argmax steps; keeping the chain inside one workgroup makes a step one barrier
//                    (~0.3 us) instead of one cross-workgroup hand-off (~2.5 us).  The panel is taken in sub-panels of W columns
//                    whose rows live in REGISTERS (thread t owns rows k0 + t + 512 q, q < R: R x W doubles); the updates of the
//                    earlier sub-panels reach a sub-panel when it is loaded (left-looking inside the panel: L from global memory /
//                    L2, the pivot rows U from LDS).  Rows are never moved while the panel is worked on: a row's data stays at its
//                    physical row `loc` and the (position -> loc) map travels with the register copy through the pivot exchanges;
//                    the panel's own 64 columns are permuted into place at the end and the map goes out as a list of at most
//                    128 (position, source) pairs.
//   lu_swap_trsm     one workgroup per 64-column slab right of the panel (the right-hand side included): applies the list (a
//                    gather through registers: all reads, barrier, all writes), then U12 = L11^-1 A12 (unit lower, in LDS).
//   lu_gemm          A22 -= L21 U12 on 64 x 64 tiles, v_mfma_f64_16x16x4_f64, one 32 x 32 quadrant per wave.
//
//   The back substitution reuses the Cholesky path's two launches (chol_inv_diag + chol_back_all, solver.hpp): lu_transpose_upper
//   writes U^T over the lower triangle (L is dead by then), the right-hand side column into the border row and 1 / u_ii into rd.
#ifndef SFFT_AMD_LU_HPP
#define SFFT_AMD_LU_HPP

#define LU_NB 64
#define LU_NT 512
#define LU_MAXTOUCH (2 * LU_NB)
#define LU_MAX_ROWS (LU_NT * 64)     // rows of the largest panel: 64 per thread

struct LuPerm { int count; int pad[3]; int pos[LU_MAXTOUCH]; int src[LU_MAXTOUCH]; };     // one per panel

template <int W> __device__ __forceinline__ void lu_ld_row(const double* __restrict__ p, double (&out)[W])
{
    if constexpr (W >= 2) {
#pragma unroll
        for (int jj = 0; jj < W; jj += 2) {
            const double2 v = *reinterpret_cast<const double2*>(p + jj);
            out[jj] = v.x; out[jj + 1] = v.y;
        }
    } else out[0] = p[0];
}

template <int W> __device__ __forceinline__ void lu_st_row(double* __restrict__ p, const double (&in)[W])
{
    if constexpr (W >= 2) {
#pragma unroll
        for (int jj = 0; jj < W; jj += 2) *reinterpret_cast<double2*>(p + jj) = make_double2(in[jj], in[jj + 1]);
    } else p[0] = in[0];
}

// lazy (left-looking) update of the registers' sub-panel with C earlier panel columns kb .. kb + C - 1:
// a[q][jj] -= L[row q][kb + u] * U[kb + u][c0 + jj], taken QG rows at a time (QG x C doubles of L in flight per thread)
template <int W, int R, int QG, int C>
__device__ __forceinline__ void lu_lazy_chunk(double (&a)[R][W], const int (&loc)[R], const double* __restrict__ A, int ld, int k0, int kb, int c0,
                                              const double (*Ub)[LU_NB + 1])
{
#pragma unroll
    for (int q0 = 0; q0 < R; q0 += QG) {
        double l[QG][C];
#pragma unroll
        for (int q = 0; q < QG; ++q) lu_ld_row<C>(A + (size_t)loc[q0 + q] * ld + k0 + kb, l[q]);
#pragma unroll
        for (int u = 0; u < C; ++u) {
            double ub[W];
#pragma unroll
            for (int jj = 0; jj < W; ++jj) ub[jj] = Ub[kb + u][c0 + jj];
#pragma unroll
            for (int q = 0; q < QG; ++q)
#pragma unroll
                for (int jj = 0; jj < W; ++jj) a[q0 + q][jj] = fma(-l[q][u], ub[jj], a[q0 + q][jj]);
        }
    }
}

template <int W, int R>
__global__ void __launch_bounds__(LU_NT) lu_panel(double* __restrict__ A, int ld, int n, int k0, int nb, LuPerm* __restrict__ perm,
                                                  int* __restrict__ status)
{
    constexpr int NW = LU_NT / 64;
    __shared__ double Ub[LU_NB][LU_NB + 1];         // pivot rows of the panel: Ub[i][c] = U[k0 + i][k0 + c], c >= i
    __shared__ double Lp[W][LU_NB + 1];             // L entries of the newest W pivot rows (panel columns < c0 + W)
    __shared__ double Tt[W][LU_NB + 1];             // their entries right of the sub-panel with the earlier sub-panels' updates applied
    __shared__ __attribute__((aligned(16))) double cdat[2][NW][W < 2 ? 2 : W];      // candidate row of each wave
    __shared__ __attribute__((aligned(16))) double ddat[2][W < 2 ? 2 : W];          // the row that sits at the pivot position
    __shared__ double cval[2][NW];
    __shared__ int crow[2][NW], cloc[2][NW], dloc[2];
    __shared__ int ploc[LU_NB];                     // physical row that holds pivot row k0 + i
    __shared__ int tcount, tpos[LU_MAXTOUCH], tsrc[LU_MAXTOUCH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // rows of this thread: position pos(q) = k0 + tid + LU_NT q (constant); loc[q] = physical row whose data sits at that position now
    double a[R][W];
    int loc[R];
#pragma unroll
    for (int q = 0; q < R; ++q) loc[q] = min(k0 + tid + LU_NT * q, n - 1);      // (slots past the last row point at a valid row and stay masked)
    if (tid == 0) tcount = 0;

    for (int c0 = 0; c0 < nb; c0 += W) {
        const bool full = (c0 + W <= nb);
        // first slot that holds rows at or below the sub-panel's first pivot position (positions k0 .. k0 + 63 are slot 0 of threads 0 .. 63)
        // ---- (a) load the sub-panel's columns of every row at or below position k0 + c0 ------------------------------------
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const double* __restrict__ src = A + (size_t)loc[q] * ld + k0 + c0;
            if (full) lu_ld_row<W>(src, a[q]);
            else {
#pragma unroll
                for (int jj = 0; jj < W; ++jj) a[q][jj] = src[min(jj, nb - 1 - c0)];
            }
        }
        // ---- (b) the earlier sub-panels' updates (rows above position k0 + c0 are pivot rows: untouched here, masked below) -------
        if (c0 > 0) {
            // QG x C doubles of L in flight per thread (<= 32): 16-byte loads wherever the chunk allows
            constexpr int QG = R < 16 ? R : 16;
            constexpr int C = (QG * 8 <= 32) ? 8 : (QG * 4 <= 32) ? 4 : 2;
            int kb = 0;
            for (; kb + C <= c0; kb += C) lu_lazy_chunk<W, R, QG, C>(a, loc, A, ld, k0, kb, c0, Ub);
            for (; kb < c0; ++kb) lu_lazy_chunk<W, R, QG, 1>(a, loc, A, ld, k0, kb, c0, Ub);
        }
        // ---- (c) factor the W columns: one barrier per column -----------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (c0 + j < nb) {                                  // (workgroup uniform)
                const int kd = k0 + c0 + j;                     // the position this column's pivot row goes to
                const int buf = j & 1;
                double bv = -1.0; int br = 0x7fffffff;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int pos = k0 + tid + LU_NT * q;
                    const double v = fabs(a[q][j]);
                    if (pos >= kd && pos < n && v > bv) { bv = v; br = pos; }          // q ascending: ties keep the first position
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double ov = __shfl_xor(bv, off); const int orow = __shfl_xor(br, off);
                    if (ov > bv || (ov == bv && orow < br)) { bv = ov; br = orow; }
                }
#pragma unroll
                for (int q = 0; q < R; ++q)
                    if (k0 + tid + LU_NT * q == br) {           // the wave's candidate: its whole sub-panel row and its physical row
#pragma unroll
                        for (int jj = 0; jj < W; ++jj) cdat[buf][wv][jj] = a[q][jj];
                        cloc[buf][wv] = loc[q];
                    }
                if (lane == 0) { cval[buf][wv] = bv; crow[buf][wv] = br; }
                if (tid == c0 + j) {                            // the row that sits at the pivot position (slot 0 of thread c0 + j)
#pragma unroll
                    for (int jj = 0; jj < W; ++jj) ddat[buf][jj] = a[0][jj];
                    dloc[buf] = loc[0];
                }
                __syncthreads();
                double gv = -1.0; int gr = 0x7fffffff, gw = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const double v = cval[buf][w]; const int r = crow[buf][w];
                    if (v > gv || (v == gv && r < gr)) { gv = v; gr = r; gw = w; }
                }
                if (gv > 0.0) {
                    double prow[W];
#pragma unroll
                    for (int jj = 0; jj < W; ++jj) prow[jj] = cdat[buf][gw][jj];
                    const int pl = cloc[buf][gw];
                    if (gr != kd) {                              // the row from the pivot position goes where the pivot row came from
#pragma unroll
                        for (int q = 0; q < R; ++q)
                            if (k0 + tid + LU_NT * q == gr) {
#pragma unroll
                                for (int jj = 0; jj < W; ++jj) a[q][jj] = ddat[buf][jj];
                                loc[q] = dloc[buf];
                            }
                    }
                    if (tid == c0 + j) {
#pragma unroll
                        for (int jj = 0; jj < W; ++jj) a[0][jj] = prow[jj];
                        loc[0] = pl;
                    }
                    const double rp = 1.0 / prow[j];             // (dgetf2 scales by the reciprocal too)
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        const int pos = k0 + tid + LU_NT * q;
                        if (pos > kd && pos < n) {
                            const double l = a[q][j] * rp;
                            a[q][j] = l;
#pragma unroll
                            for (int jj = 0; jj < W; ++jj)          // (constant trip count: `jj = j + 1` stays a loop and keeps `a` in scratch)
                                if (jj > j) a[q][jj] = fma(-l, prow[jj], a[q][jj]);
                        }
                    }
                    if (tid == 0) ploc[c0 + j] = pl;
                    if (tid >= j && tid < W) Ub[c0 + j][c0 + tid] = cdat[buf][gw][tid];
                } else {                                         // no nonzero (or only NaN) entries left in this column: singular
                    if (tid == 0) atomicOr(status, 2);
                    if (tid == c0 + j) {
                        ploc[c0 + j] = loc[0];
#pragma unroll
                        for (int jj = 0; jj < W; ++jj)
                            if (jj >= j) Ub[c0 + j][c0 + jj] = a[0][jj];
                    }
                }
            }
        }
        // ---- (d) the sub-panel goes back to its physical rows ------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int pos = k0 + tid + LU_NT * q;
            if (pos >= k0 + c0 && pos < n) {
                double* __restrict__ dst = A + (size_t)loc[q] * ld + k0 + c0;
                if (full) lu_st_row<W>(dst, a[q]);
                else {
#pragma unroll
                    for (int jj = 0; jj < W; ++jj) if (c0 + jj < nb) dst[jj] = a[q][jj];
                }
            }
        }
        __syncthreads();                // the stores are visible to the workgroup; ploc / Ub of this sub-panel are complete
        // ---- (f) the new pivot rows right of the sub-panel become U rows now (the later sub-panels need them in (b)) -------------
        const int cr0 = c0 + W, ncr = nb - cr0;
        if (ncr > 0) {
            for (int e = tid; e < W * cr0; e += LU_NT) {
                const int i = e / cr0, kk = e - i * cr0;
                Lp[i][kk] = A[(size_t)ploc[c0 + i] * ld + k0 + kk];
            }
            __syncthreads();
            for (int e = tid; e < W * ncr; e += LU_NT) {
                const int i = e / ncr, c = cr0 + (e - i * ncr);
                double t = A[(size_t)ploc[c0 + i] * ld + k0 + c];
                for (int kk = 0; kk < c0; ++kk) t = fma(-Lp[i][kk], Ub[kk][c], t);
                Tt[i][c] = t;
            }
            __syncthreads();
            if (tid < ncr) {
                const int c = cr0 + tid;
                double u[W];
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    double t = Tt[i][c];
#pragma unroll
                    for (int i2 = 0; i2 < W; ++i2)
                        if (i2 < i) t = fma(-Lp[i][c0 + i2], u[i2], t);
                    u[i] = t;
                    Ub[c0 + i][c] = t;
                    A[(size_t)ploc[c0 + i] * ld + k0 + c] = t;
                }
            }
            __syncthreads();
        }
    }

    // ---- the panel's own columns are permuted into place; the list goes out for the slabs right of the panel ----------------------
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int pos = k0 + tid + LU_NT * q;
        if (pos < n && loc[q] != pos) {
            const int e = atomicAdd(&tcount, 1);
            if (e < LU_MAXTOUCH) { tpos[e] = pos; tsrc[e] = loc[q]; }
        }
    }
    __syncthreads();
    const int cnt = min(tcount, LU_MAXTOUCH);
    {
        constexpr int PER = LU_MAXTOUCH * LU_NB / LU_NT;        // 16
        double tmp[PER];
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = tid + LU_NT * it, e = idx >> 6, c = idx & 63;
            tmp[it] = (e < cnt && c < nb) ? A[(size_t)tsrc[e] * ld + k0 + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = tid + LU_NT * it, e = idx >> 6, c = idx & 63;
            if (e < cnt && c < nb) A[(size_t)tpos[e] * ld + k0 + c] = tmp[it];
        }
    }
    if (tid == 0) perm->count = cnt;
    if (tid < cnt) { perm->pos[tid] = tpos[tid]; perm->src[tid] = tsrc[tid]; }
}

// One workgroup per 64-column slab right of the panel: rows into place, then U12 = L11^-1 A12.
__global__ void __launch_bounds__(256) lu_swap_trsm(double* __restrict__ A, int ld, int n, int k0, int nb, const LuPerm* __restrict__ perm)
{
    __shared__ double Ls[LU_NB][LU_NB + 1];
    __shared__ double Bs[LU_NB][LU_NB + 1];
    __shared__ int spos[LU_MAXTOUCH], ssrc[LU_MAXTOUCH];
    const int tid = threadIdx.x;
    const int cs = k0 + nb + LU_NB * (int)blockIdx.x, wc = min(LU_NB, n + 1 - cs);
    const int cnt = perm->count;
    if (tid < LU_MAXTOUCH) { spos[tid] = tid < cnt ? perm->pos[tid] : 0; ssrc[tid] = tid < cnt ? perm->src[tid] : 0; }
    __syncthreads();
    {
        constexpr int PER = LU_MAXTOUCH * LU_NB / 256;          // 32
        double tmp[PER];
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = tid + 256 * it, e = idx >> 6, c = idx & 63;
            tmp[it] = (e < cnt && c < wc) ? A[(size_t)ssrc[e] * ld + cs + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = tid + 256 * it, e = idx >> 6, c = idx & 63;
            if (e < cnt && c < wc) A[(size_t)spos[e] * ld + cs + c] = tmp[it];
        }
    }
    __syncthreads();
    {
        double lv[16], bv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it, r = e >> 6, c = e & 63;
            lv[it] = A[(size_t)(k0 + min(r, nb - 1)) * ld + k0 + min(c, nb - 1)];
            bv[it] = A[(size_t)(k0 + min(r, nb - 1)) * ld + cs + min(c, wc - 1)];
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it, r = e >> 6, c = e & 63;
            Ls[r][c] = (r < nb && c < r) ? lv[it] : 0.0;
            Bs[r][c] = (r < nb && c < wc) ? bv[it] : 0.0;
        }
    }
    const int c = tid & 63, rg = tid >> 6;
    for (int i = 0; i + 1 < nb; ++i) {
        __syncthreads();
        const double bi = Bs[i][c];
        for (int r = i + 1 + rg; r < nb; r += 4) Bs[r][c] = fma(-Ls[r][i], bi, Bs[r][c]);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, r = e >> 6, cc = e & 63;
        if (r < nb && cc < wc) A[(size_t)(k0 + r) * ld + cs + cc] = Bs[r][cc];
    }
}

// A22 -= L21 U12: tile (blockIdx.y, blockIdx.x) of 64 x 64 below / right of the panel; K = nb <= 64.
#define LU_GS 66
__global__ void __launch_bounds__(256) lu_gemm(double* __restrict__ A, int ld, int n, int k0, int nb)
{
    __shared__ double Ls[LU_NB][LU_GS];             // [row][k]
    __shared__ double Us[LU_NB][LU_NB + 16];        // [k][column]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, lk = lane >> 4;
    const int wr = wv >> 1, wc = wv & 1;
    const int r0 = k0 + nb + LU_NB * (int)blockIdx.y, c0 = k0 + nb + LU_NB * (int)blockIdx.x;
    const int nr = min(LU_NB, n - r0), nc = min(LU_NB, n + 1 - c0);
    {
        double2 lv[8], uv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = tid + 256 * it, r = e >> 5, c = 2 * (e & 31);
            // (k0 + c + 1 < ld and c0 + c + 1 < ld always hold inside a row; rows are clamped, then masked)
            lv[it] = *reinterpret_cast<const double2*>(A + (size_t)(r0 + min(r, nr - 1)) * ld + k0 + min(c, (nb - 1) & ~1));
            uv[it] = *reinterpret_cast<const double2*>(A + (size_t)(k0 + min(r, nb - 1)) * ld + c0 + min(c, (nc - 1) & ~1));
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = tid + 256 * it, r = e >> 5, c = 2 * (e & 31);
            Ls[r][c] = (r < nr && c < nb) ? lv[it].x : 0.0;
            Ls[r][c + 1] = (r < nr && c + 1 < nb) ? lv[it].y : 0.0;
            Us[r][c] = (r < nb && c < nc) ? uv[it].x : 0.0;
            Us[r][c + 1] = (r < nb && c + 1 < nc) ? uv[it].y : 0.0;
        }
    }
    __syncthreads();
    d4s acc[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[it][jt] = (d4s){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int ks = 0; ks < LU_NB / 4; ++ks) {
        double av[2], bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            av[t] = Ls[32 * wr + 16 * t + ln][4 * ks + lk];
            bv[t] = Us[4 * ks + lk][32 * wc + 16 * t + ln];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) acc[it][jt] = mfma16(av[it], bv[jt], acc[it][jt]);
    }
    // accumulator element q of lane (ln, lk): row 4 lk + q?  -- as chol_syrk: row = lk + 4 q, column = ln
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        double oldv[2][4];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 32 * wr + 16 * it + lk + 4 * q, j = 32 * wc + 16 * jt + ln;
                oldv[jt][q] = (i < nr && j < nc) ? A[(size_t)(r0 + i) * ld + c0 + j] : 0.0;
            }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 32 * wr + 16 * it + lk + 4 * q, j = 32 * wc + 16 * jt + ln;
                if (i < nr && j < nc) A[(size_t)(r0 + i) * ld + c0 + j] = oldv[jt][q] - acc[it][jt][q];
            }
    }
}

// After the last panel: U^T over the lower triangle (tile (bi, bj), bj >= bi, read from the upper triangle and written
// transposed), the right-hand side column y = L^-1 P b into the border row n, rd[i] = 1 / u_ii -- the layout chol_inv_diag and
// chol_back_all expect of a Cholesky factor (x_b = W_b^T (y_b - sum_{c > b} L_cb^T x_c) with L_cb = U_bc^T).
__global__ void __launch_bounds__(256) lu_transpose_upper(double* __restrict__ A, int ld, int n, double* __restrict__ rd, int* __restrict__ status)
{
    __shared__ double T[LU_NB][LU_NB + 1];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int tid = threadIdx.x, r0 = bi * LU_NB, c0 = bj * LU_NB;
    const int nr = min(LU_NB, n - r0), nc = min(LU_NB, n + 1 - c0);      // column n (the right-hand side) included
    double v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, r = e >> 6, c = e & 63;
        v[it] = A[(size_t)(r0 + min(r, nr - 1)) * ld + c0 + min(c, nc - 1)];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, r = e >> 6, c = e & 63;
        T[r][c] = v[it];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, r = e >> 6, c = e & 63;      // writes element (column c0 + r of the source, row r0 + c of the source) -> A[c0 + r][r0 + c]
        if (r < nc && c < nr && (bj > bi ? true : r > c)) A[(size_t)(c0 + r) * ld + r0 + c] = T[c][r];
    }
    if (bi == bj && tid < nr) {
        const double d = T[tid][tid];
        if (!(fabs(d) > 0.0)) atomicOr(status, 2);
        rd[r0 + tid] = 1.0 / d;
    }
}

#endif
