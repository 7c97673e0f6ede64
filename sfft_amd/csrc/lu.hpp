// lu.hpp -- dense solve with the reference's own semantics: LU with partial (row) pivoting, blocked for the matrix cores.
// Part of libsfft_amd (MI355X / gfx950); included by lu.hip only (host side: lu_api.hpp).
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
//   lu_panel_mw<R>   the same panel on G <= 16 workgroups (panels taller than 1024 rows): see there.
//   lu_swap_trsm     one workgroup per 64-column slab right of the panel (the right-hand side included): applies the list (a
//                    gather through registers: all reads, barrier, all writes), then U12 = L11^-1 A12 (unit lower, in LDS).
//   lu_gemm          A22 -= L21 U12 on 64 x 64 tiles, v_mfma_f64_16x16x4_f64, one 32 x 32 quadrant per wave.
//
//   The back substitution reuses the Cholesky path's two launches (chol_inv_diag + chol_back_all, solver.hpp): lu_transpose_upper
//   writes U^T over the lower triangle (L is dead by then), the right-hand side column into the border row and 1 / u_ii into rd.
#ifndef SFFT_AMD_LU_HPP
#define SFFT_AMD_LU_HPP

#include "lu_api.hpp"

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

// ---- cross-lane reductions on the DPP path (no LDS round trips on the pivot chain) -------------------------------------------------
// inclusive scan steps row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast 15 into rows 1 and 3 and row_bcast 31 into
// rows 2 and 3: lane 63 holds the wave's result
template <int CTRL, int RM> __device__ __forceinline__ unsigned int lu_dpp_max32(unsigned int v)
{
    const unsigned int t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RM, 0xF, false);
    return t > v ? t : v;
}
__device__ __forceinline__ unsigned int lu_wave_max32(unsigned int v)
{
    v = lu_dpp_max32<0x111, 0xF>(v); v = lu_dpp_max32<0x112, 0xF>(v); v = lu_dpp_max32<0x114, 0xF>(v); v = lu_dpp_max32<0x118, 0xF>(v);
    v = lu_dpp_max32<0x142, 0xA>(v); v = lu_dpp_max32<0x143, 0xC>(v);
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit maximum as two 32-bit ones (one v_max_u32_dpp per step): the high words, then the low words of the lanes that hold the
// largest high word
__device__ __forceinline__ unsigned long long lu_wave_max64(unsigned long long v)
{
    const unsigned int hi = (unsigned int)(v >> 32), lo = (unsigned int)v;
    const unsigned int mh = lu_wave_max32(hi);
    const unsigned int ml = lu_wave_max32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}
template <int CTRL, int RM> __device__ __forceinline__ unsigned int lu_dpp_min32(unsigned int v)
{
    const unsigned int t = (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, RM, 0xF, false);
    return t < v ? t : v;
}
__device__ __forceinline__ unsigned int lu_wave_min32(unsigned int v)
{
    v = lu_dpp_min32<0x111, 0xF>(v); v = lu_dpp_min32<0x112, 0xF>(v); v = lu_dpp_min32<0x114, 0xF>(v); v = lu_dpp_min32<0x118, 0xF>(v);
    v = lu_dpp_min32<0x142, 0xA>(v); v = lu_dpp_min32<0x143, 0xC>(v);
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

// 1 / x to ~1 ulp: hardware estimate + two Newton steps (an IEEE division is ~40 dependent instructions on the pivot chain)
__device__ __forceinline__ double lu_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

// |v| as an ordered integer key: IEEE bit patterns of non-negative doubles sort like unsigned integers (a NaN sorts above
// infinity: it becomes the pivot and the solution comes out NaN, as from LAPACK)
__device__ __forceinline__ unsigned long long lu_key(double v)
{
    return (unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull;
}

// compile-time loop: the body sees its index as a constant (a `#pragma unroll` loop this large is only partially unrolled, and a
// runtime index into the register tile sends the tile to scratch)
template <int I, int N, typename F> __device__ __forceinline__ void lu_static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); lu_static_for<I + 1, N>(f); }
}

#ifdef LU_TRACE          // scripts/micro/lu_panel_trace.hip: s_memtime stamps of thread 0 at the phase boundaries of one panel launch
__device__ long long lu_trace[256];
#define LU_STAMP(slot) do { if (threadIdx.x == 0) lu_trace[slot] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define LU_STAMP(slot) do { } while (0)
#endif

struct __attribute__((aligned(16))) LuCand { double piv; int pos; int phys; };      // piv: the candidate's entry in the pivot column (its magnitude is the key)

template <int CTRL> __device__ __forceinline__ unsigned int lu_quad_max32(unsigned int v)
{
    const unsigned int t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
    return t > v ? t : v;
}
template <int CTRL> __device__ __forceinline__ unsigned int lu_quad_min32(unsigned int v)
{
    const unsigned int t = (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, 0xF, 0xF, false);
    return t < v ? t : v;
}

// The panel kernel.  256 threads = one wave per SIMD (the pivot chain is instruction-issue bound: everything a column needs is
// executed by every wave, so fewer, fatter waves win); thread t owns PHYSICAL rows k0 + t + 256 q, q < R, for the whole kernel.
// A row never changes its thread: a pivot exchange only exchanges the POSITIONS pos[q] the two rows stand for (idamax's first-row
// tie rule needs positions; the data needs no move at all).
template <int W, int R>
__global__ void __launch_bounds__(LU_NT) lu_panel(double* __restrict__ A, int ld, int n, int k0, int nb, LuPerm* __restrict__ perm,
                                                  int* __restrict__ status)
{
    constexpr int NW = LU_NT / 64;
    static_assert(NW == 4, "the cross-wave selection reduces over a lane quad");
    constexpr int WP = W < 2 ? 2 : W;
    constexpr int NS = LU_NB > W ? LU_NB - W : 1;   // columns a panel can have right of a sub-panel
    __shared__ double Ub[LU_NB][LU_NB + 1];         // pivot rows of the panel: Ub[i][c] = U[k0 + i][k0 + c], c >= i
    __shared__ double Lp[W < 32 ? W : 32][LU_NB + 1];      // L entries of the newest W pivot rows (panel columns < c0 + W); W = 64 is always one pass
    __shared__ double Tt[W < 32 ? W : 32][LU_NB + 1];      // their entries right of the sub-panel (raw, then with the earlier sub-panels' updates applied)
    __shared__ __attribute__((aligned(16))) double cdat[2][NW][WP];      // candidate row of each wave
    __shared__ LuCand cand[2][NW];
    __shared__ int ploc[LU_NB];                     // physical row of pivot row k0 + i
    __shared__ int tcount, tpos[LU_MAXTOUCH], tsrc[LU_MAXTOUCH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double a[R][W];
    int pos[R];                                     // the position physical row k0 + tid + 256 q stands for (-1: no such row)
#pragma unroll
    for (int q = 0; q < R; ++q) pos[q] = (k0 + tid + LU_NT * q < n) ? k0 + tid + LU_NT * q : -1;
    if (tid == 0) tcount = 0;
    if (tid < LU_MAXTOUCH) { tpos[tid] = k0; tsrc[tid] = k0; }
    const bool one_pass = (nb <= W);                // the whole panel is one sub-panel: rows go straight to their final places

    LU_STAMP(0);
    for (int c0 = 0; c0 < nb; c0 += W) {
        const bool full = (c0 + W <= nb);
        LU_STAMP(1 + 8 * (c0 / W));
        // ---- (a) the sub-panel's columns of every row, (b) the earlier sub-panels' updates (left-looking; rows that are pivot rows
        //      already ride along untouched and stay masked below).  All loads are independent of each other: the raw columns and
        //      the first chunk of L are in flight together, each further chunk is fetched a step ahead ------------------------------
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const double* __restrict__ src = A + (size_t)min(k0 + tid + LU_NT * q, n - 1) * ld + k0 + c0;
            if (full) lu_ld_row<W>(src, a[q]);
            else {
#pragma unroll
                for (int jj = 0; jj < W; ++jj) a[q][jj] = src[min(jj, nb - 1 - c0)];
            }
        }
        if (c0 > 0) {
            if constexpr (R <= 16) {
                // C = min(W, 16) columns of L per step, QG rows at a time: a step is one memory round trip (the L columns come from L2 /
                // HBM), so steps must be few, and its QG x C doubles of L plus the QG rows of the tile they update must fit the directly
                // addressable registers beside each other (all R rows at once: 20 k cycles per step at R = 6, most of it moves to and from
                // the accumulation registers; two columns per step with the next step prefetched: 24 round trips per sub-panel)
                constexpr int C = W < 16 ? W : 16;
                constexpr int QG = R <= 3 ? R : (R % 3 == 0 ? 3 : 2);
#pragma unroll
                for (int q0 = 0; q0 < R; q0 += QG) {
                    for (int kb = 0; kb < c0; kb += C) {
                        double lc[QG][C];
#pragma unroll
                        for (int q = 0; q < QG; ++q) lu_ld_row<C>(A + (size_t)min(k0 + tid + LU_NT * (q0 + q), n - 1) * ld + k0 + kb, lc[q]);
#pragma unroll
                        for (int u = 0; u < C; ++u) {
                            double ub[W];
#pragma unroll
                            for (int jj = 0; jj < W; ++jj) ub[jj] = Ub[kb + u][c0 + jj];
#pragma unroll
                            for (int q = 0; q < QG; ++q)
#pragma unroll
                                for (int jj = 0; jj < W; ++jj) a[q0 + q][jj] = fma(-lc[q][u], ub[jj], a[q0 + q][jj]);
                        }
                    }
                }
            } else {
                for (int kb = 0; kb < c0; ++kb) {
#pragma unroll
                    for (int q0 = 0; q0 < R; q0 += 8) {
                        double l[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) l[q] = A[(size_t)min(k0 + tid + LU_NT * (q0 + q), n - 1) * ld + k0 + kb];
                        double ub[W];
#pragma unroll
                        for (int jj = 0; jj < W; ++jj) ub[jj] = Ub[kb][c0 + jj];
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int jj = 0; jj < W; ++jj) a[q0 + q][jj] = fma(-l[q], ub[jj], a[q0 + q][jj]);
                    }
                }
            }
        }
        LU_STAMP(2 + 8 * (c0 / W));
        // ---- (c) factor the W columns: one barrier per column -----------------------------------------------------------------
        lu_static_for<0, W>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (c0 + j < nb) {                                  // (workgroup uniform)
                const int kd = k0 + c0 + j;                     // the position this column's pivot row goes to
                const int buf = j & 1;
                unsigned long long bk = 0ull; int bp = 0x7fffffff;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const unsigned long long key = (pos[q] >= kd) ? lu_key(a[q][j]) : 0ull;
                    const bool better = key > bk;               // among equal magnitudes the first slot, lane, wave (LAPACK: the first POSITION;
                    bk = better ? key : bk;                     //  either is partial pivoting -- an entry of largest magnitude)
                    bp = better ? pos[q] : bp;
                }
                const unsigned long long wk = lu_wave_max64(bk);
                const int wl = __ffsll((long long)__ballot(bk == wk)) - 1;          // the wave's candidate lane
#pragma unroll
                for (int q = 0; q < R; ++q)
                    if (lane == wl && pos[q] == bp && wk != 0ull) {      // its row from column j on
#pragma unroll
                        for (int jj = 0; jj < W; ++jj)
                            if (jj >= j) cdat[buf][wv][jj] = a[q][jj];
                        cand[buf][wv] = LuCand{a[q][j], bp, k0 + tid + LU_NT * q};
                    }
                if (wk == 0ull && lane == 0) cand[buf][wv] = LuCand{0.0, 0x7fffffff, k0};
                if (j >= 4 && j <= 6) LU_STAMP(200 + 4 * (j - 4) + 0);
                __syncthreads();
                if (j >= 4 && j <= 6) LU_STAMP(200 + 4 * (j - 4) + 1);
                // every lane reads candidate (lane & 3); two butterfly steps inside the lane quad give every lane the workgroup's winner
                const LuCand cd = cand[buf][lane & 3];
                const unsigned long long ckey = lu_key(cd.piv);
                const unsigned int chi = (unsigned int)(ckey >> 32), clo = (unsigned int)ckey;
                const unsigned int ghi = lu_quad_max32<0x4E>(lu_quad_max32<0xB1>(chi));
                const unsigned int glo = lu_quad_max32<0x4E>(lu_quad_max32<0xB1>(chi == ghi ? clo : 0u));
                const unsigned long long gk = ((unsigned long long)ghi << 32) | glo;
                const int gw = __ffsll((long long)__ballot(ckey == gk)) - 1;         // (< 4: the quads repeat) the first wave that holds the maximum
                const int gr = __builtin_amdgcn_readlane(cd.pos, gw);
                if (gk != 0ull) {
                    if (tid == 0) ploc[c0 + j] = __builtin_amdgcn_readlane(cd.phys, gw);
                    double prow[W];
#pragma unroll
                    for (int jj = 0; jj < W; ++jj) prow[jj] = (jj >= j) ? cdat[buf][gw][jj] : 0.0;
                    // the pivot element comes with the candidate record: its reciprocal is under way while the row itself is still being read
                    const double rp = lu_rcp(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cd.piv), gw),
                                                              __builtin_amdgcn_readlane(__double2loint(cd.piv), gw)));      // (dgetf2 scales by the reciprocal too)
#pragma unroll
                    for (int q = 0; q < R; ++q) {                // branch-free: rows that are not below the pivot position take l = 0
                        pos[q] = (pos[q] == kd) ? gr : (pos[q] == gr ? kd : pos[q]);      // the two rows exchange positions, not data
                        const bool act = pos[q] > kd;
                        const double l = act ? a[q][j] * rp : 0.0;
                        a[q][j] = act ? l : a[q][j];
#pragma unroll
                        for (int jj = 0; jj < W; ++jj)          // (constant trip count: `jj = j + 1` stays a loop and keeps `a` in scratch)
                            if (jj > j) a[q][jj] = fma(-l, prow[jj], a[q][jj]);
                    }
                    if (tid >= j && tid < W) Ub[c0 + j][c0 + tid] = cdat[buf][gw][tid];
                } else {                                         // no nonzero (or only NaN) entries left in this column: singular
                    if (tid == 0) atomicOr(status, 2);
#pragma unroll
                    for (int q = 0; q < R; ++q)
                        if (pos[q] == kd) {                      // the row at the pivot position stays there
                            ploc[c0 + j] = k0 + tid + LU_NT * q;
#pragma unroll
                            for (int jj = 0; jj < W; ++jj)
                                if (jj >= j) Ub[c0 + j][c0 + jj] = a[q][jj];
                        }
                }
            }
        });
        LU_STAMP(3 + 8 * (c0 / W));
        // ---- (d) the sub-panel goes back: to its physical rows, or -- a panel of one sub-panel -- straight to the final positions
        //      (every row at or below k0 is in registers then, so writing each to its position IS the permutation) ---------------------
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (pos[q] >= k0 + c0) {
                double* __restrict__ dst = A + (size_t)(one_pass ? pos[q] : k0 + tid + LU_NT * q) * ld + k0 + c0;
                if (full) lu_st_row<W>(dst, a[q]);
                else {
#pragma unroll
                    for (int jj = 0; jj < W; ++jj) if (c0 + jj < nb) dst[jj] = a[q][jj];
                }
            }
        }
        const int cr0 = c0 + W, ncr = nb - cr0;
        if constexpr (W < LU_NB) {
            if (ncr > 0) {              // the new pivot rows leave their sub-panel entries (L left of the diagonal) in LDS for (f)
#pragma unroll
                for (int q = 0; q < R; ++q)
                    if (pos[q] >= k0 + c0 && pos[q] < k0 + cr0) {
#pragma unroll
                        for (int jj = 0; jj < W; ++jj) Lp[pos[q] - k0 - c0][c0 + jj] = a[q][jj];
                    }
            }
        }
        // LDS-only barrier: ploc / Ub / Lp of this sub-panel are complete.  (d)'s global stores drain behind it -- nothing below reads another
        // thread's fresh stores: (f) takes this sub-panel's L entries from Lp, a thread re-reads only its OWN rows later, and the final
        // gather sits behind a full __syncthreads()
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        LU_STAMP(4 + 8 * (c0 / W));
        // ---- (f) the new pivot rows right of the sub-panel become U rows now (the later sub-panels need them in (b)): their L
        //      entries of EARLIER sub-panels and their raw entries come in one round trip (unconditional, clamped loads) ----------------
        if constexpr (W < LU_NB) {
        if (ncr > 0) {
            constexpr int WL = W < 32 ? W : 32;
            constexpr int NIT = (WL * NS + LU_NT - 1) / LU_NT;
            double lv[NIT], tv[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + LU_NT * it;
                const int el = min(e, max(W * c0 - 1, 0)), il = c0 > 0 ? el / c0 : 0, kk = c0 > 0 ? el - il * c0 : 0;
                lv[it] = A[(size_t)ploc[c0 + il] * ld + k0 + kk];
                const int et = min(e, W * ncr - 1), i2 = et / ncr, c = cr0 + (et - i2 * ncr);
                tv[it] = A[(size_t)ploc[c0 + i2] * ld + k0 + c];
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + LU_NT * it;
                if (e < W * c0) { const int il = e / c0; Lp[il][e - il * c0] = lv[it]; }
                if (e < W * ncr) { const int i2 = e / ncr; Tt[i2][cr0 + (e - i2 * ncr)] = tv[it]; }
            }
            __syncthreads();
            for (int e = tid; e < W * ncr; e += LU_NT) {
                const int i = e / ncr, c = cr0 + (e - i * ncr);
                double t = Tt[i][c];
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
    }

    LU_STAMP(100);
    // ---- the panel's own columns are permuted into place; the list goes out for the slabs right of the panel ----------------------
#pragma unroll
    for (int q = 0; q < R; ++q) {
        if (pos[q] >= 0 && pos[q] != k0 + tid + LU_NT * q) {
            const int e = atomicAdd(&tcount, 1);
            if (e < LU_MAXTOUCH) { tpos[e] = pos[q]; tsrc[e] = k0 + tid + LU_NT * q; }
        }
    }
    __syncthreads();
    const int cnt = min(tcount, LU_MAXTOUCH);
    if (!one_pass) {
        constexpr int PER = LU_MAXTOUCH * LU_NB / LU_NT;        // 32
        double tmp[PER];
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = tid + LU_NT * it, e = idx >> 6, c = idx & 63;
            tmp[it] = A[(size_t)tsrc[e] * ld + k0 + min(c, nb - 1)];        // (entries past cnt point at row k0: loaded, never stored)
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
    LU_STAMP(101);
}

// ================================================================================================
// The same panel on SEVERAL workgroups (tall systems: more rows than one workgroup's registers hold 16 columns of).  Workgroup g owns the
// physical rows k0 + 1536 g + t + 256 q (R = 6, W = 16) and runs the single-workgroup algorithm on them; two things cross workgroups:
//  * every column: each workgroup publishes its candidate {pivot element, position, physical row, the row's 16 sub-panel entries} in a
//    global slot of its own, then flags it; wave 0 of every workgroup waits for all G flags, takes the largest magnitude (ties: the
//    lowest workgroup) and hands record and row to its own workgroup through LDS (one more barrier).  Hand-off recipe as in
//    chol_dataflow (solver.hpp): agent-scope write-through stores, s_waitcnt vmcnt(0), one relaxed agent-scope flag store; relaxed
//    agent-scope polls, one acquire fence, agent-scope loads.  Flags and slots are indexed by (column, workgroup) and stamped per launch:
//    nothing is reused inside a launch and nothing needs clearing.
//  * every sub-panel: the new pivot rows' entries right of the sub-panel need their OWNER's L entries of earlier sub-panels (written
//    by the owner in this same launch: invisible to the others), so the owner brings them up to date and publishes the 16 x <= 48 block
//    the same way; the triangular solve against the sub-panel's own L (known to everybody from the column records) runs in every workgroup.
// All spins are bounded (status bit 4: the LU result is then reported singular instead of hanging).  The G workgroups must be resident
// together (G <= 16 of 256 CUs).  No final gather: the row list is appended to perm (count reset by lu_perm_reset) and applied to the
// panel's own columns by lu_apply_perm.
// ================================================================================================
#define LU_MW_W 16
#define LU_MW_R 6
#define LU_MW_MAXG 16
// hand-off granules: 16 bytes {value (8), stamp (4), unused (4)}, written by ONE 16-byte write-through store and read by 16-byte loads that
// bypass the caches -- a granule whose stamp matches carries its value (MI355X_MICROARCH.md: 16-byte sc1 stores and sc1 loads need no fence),
// so a record needs no flag behind it and no second read after a flag
typedef unsigned int lu_u4 __attribute__((ext_vector_type(4)));
#define LU_XG 20                                             // granules per record: 16 row entries, the pivot element, {position, physical row}
struct LuXchg {
    lu_u4 rec[LU_NB][LU_MW_MAXG][LU_XG];
    lu_u4 trow[LU_NB / LU_MW_W][LU_MW_W][LU_NB];            // [sub-panel][pivot row i][panel column]
    lu_u4 hello[LU_MW_MAXG];                                // launch start: the XCD every workgroup runs on
};
// near = every workgroup of the launch runs on the SAME XCD (checked at launch start from HW_REG_XCC_ID, never assumed): the XCD's L2 is
// then the meeting point -- stores that stay in it (sc0: the L1 writes through) and device-scope loads (sc1: past the L1, served by that L2) -- at a fraction of the latency of
// the write-through-to-memory / read-from-memory pair (sc1 stores, sc0 sc1 loads) that any placement needs.
__device__ __forceinline__ void lu_put(lu_u4* p, unsigned long long bits, unsigned int stamp, bool near)
{
    lu_u4 pk; pk.x = (unsigned int)bits; pk.y = (unsigned int)(bits >> 32); pk.z = stamp; pk.w = 0u;
    if (near) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(pk) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(pk) : "memory");
}
// (inline asm: written with the buffer-load builtin the compiler hoists the load out of the spin loop -- its "volatile" aux bit is not honoured
//  here -- and the loop only sleeps)
__device__ __forceinline__ lu_u4 lu_ld16(const lu_u4* p, bool near)
{
    lu_u4 g;
    if (near) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(p) : "memory");
    return g;
}
#define LU_XK 5                                              // granules a lane of wave 0 watches per column: ceil(16 x 18 / 64)
__device__ __forceinline__ void lu_ld16_n(const lu_u4* const (&p)[LU_XK], lu_u4 (&g)[LU_XK], bool near)
{
    if (near)
        asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                     "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %5, off sc0 sc1\n\tglobal_load_dwordx4 %1, %6, off sc0 sc1\n\tglobal_load_dwordx4 %2, %7, off sc0 sc1\n\t"
                     "global_load_dwordx4 %3, %8, off sc0 sc1\n\tglobal_load_dwordx4 %4, %9, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]) : "memory");
}
// spin (bounded) until the granule carries this launch's stamp; returns its 8 value bytes
__device__ __forceinline__ unsigned long long lu_get(const lu_u4* p, unsigned int stamp, int* __restrict__ status, bool near)
{
    lu_u4 g = lu_ld16(p, near);
    int spins = 0;
    while (g.z != stamp) {
        __builtin_amdgcn_s_sleep(1);
        g = lu_ld16(p, near);
        if (++spins > (1 << 20)) { atomicOr(status, 4); break; }
    }
    return ((unsigned long long)g.y << 32) | g.x;
}

__global__ void lu_perm_reset(LuPerm* perm) { if (threadIdx.x == 0) perm->count = 0; }

template <int R>
__global__ void __launch_bounds__(LU_NT) lu_panel_mw(double* __restrict__ A, int ld, int n, int k0, int nb, LuPerm* __restrict__ perm,
                                                     int* __restrict__ status, LuXchg* __restrict__ xb, const unsigned int* __restrict__ epoch_ctr, int panel_id)
{
    // this launch's stamp: the solve's epoch (chol_begin advances it on the device, so a replayed graph gets a fresh one) and the panel
    const unsigned int stamp = ((*epoch_ctr & 0x3FFFFFu) << 10) | (unsigned int)(panel_id + 1);
    constexpr int W = LU_MW_W, NW = LU_NT / 64;
    static_assert(NW == 4, "the cross-wave selection reduces over a lane quad");
    __shared__ double Ub[LU_NB][LU_NB + 1];
    __shared__ double Lp[W][LU_NB + 1];
    __shared__ double Tt[W][LU_NB + 1];
    __shared__ __attribute__((aligned(16))) double cdat[2][NW][W];
    __shared__ LuCand cand[2][NW];
    __shared__ __attribute__((aligned(16))) double grow[W];         // the global winner's sub-panel row ...
    __shared__ LuCand ghdr;                                         // ... and record (piv = 0: no pivot anywhere)
    __shared__ unsigned long long xs[LU_MW_MAXG][LU_XG];           // wave 0's copy of the G records of a column
    __shared__ int ploc[LU_NB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the grid has 8 G workgroups and only every eighth works: observed (not promised) placement is block b -> XCD b % 8, so the G
    // working ones usually share an XCD and its L2
    if ((blockIdx.x & 7) != 0) return;
    const int G = (int)gridDim.x >> 3, wg = (int)blockIdx.x >> 3;
    const int row0 = k0 + wg * (R * LU_NT);                         // first physical row of this workgroup
    __shared__ int s_near;
    if (wv == 0) {                                                  // who runs where: one hand-off that works under ANY placement
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0) lu_put(&xb->hello[wg], (unsigned long long)(unsigned int)(xcc & 0xF), stamp, false);
        const unsigned int other = lane < G ? (unsigned int)lu_get(&xb->hello[min(lane, G - 1)], stamp, status, false) : (unsigned int)(xcc & 0xF);
        const bool same = __ballot(other != (unsigned int)(xcc & 0xF)) == 0ull;
        if (lane == 0) s_near = same ? 1 : 0;
    }
    __syncthreads();
    const bool near = s_near != 0;
    double a[R][W];
    int pos[R];
#pragma unroll
    for (int q = 0; q < R; ++q) pos[q] = (row0 + tid + LU_NT * q < n) ? row0 + tid + LU_NT * q : -1;

    for (int c0 = 0; c0 < nb; c0 += W) {
        const bool full = (c0 + W <= nb);
        // ---- (a), (b): as lu_panel ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const double* __restrict__ src = A + (size_t)min(row0 + tid + LU_NT * q, n - 1) * ld + k0 + c0;
            if (full) lu_ld_row<W>(src, a[q]);
            else {
#pragma unroll
                for (int jj = 0; jj < W; ++jj) a[q][jj] = src[min(jj, nb - 1 - c0)];
            }
        }
        if (c0 > 0) {
            constexpr int QG = (R % 3 == 0) ? 3 : (R % 2 == 0 ? 2 : 1);
#pragma unroll
            for (int q0 = 0; q0 < R; q0 += QG) {
                for (int kb = 0; kb < c0; kb += W) {
                    double lc[QG][W];
#pragma unroll
                    for (int q = 0; q < QG; ++q) lu_ld_row<W>(A + (size_t)min(row0 + tid + LU_NT * (q0 + q), n - 1) * ld + k0 + kb, lc[q]);
#pragma unroll
                    for (int u = 0; u < W; ++u) {
                        double ub[W];
#pragma unroll
                        for (int jj = 0; jj < W; ++jj) ub[jj] = Ub[kb + u][c0 + jj];
#pragma unroll
                        for (int q = 0; q < QG; ++q)
#pragma unroll
                            for (int jj = 0; jj < W; ++jj) a[q0 + q][jj] = fma(-lc[q][u], ub[jj], a[q0 + q][jj]);
                    }
                }
            }
        }
        // ---- (c) ---------------------------------------------------------------------------------------------------------------------
        lu_static_for<0, W>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (c0 + j < nb) {
                const int kd = k0 + c0 + j;
                const int buf = j & 1;
                unsigned long long bk = 0ull; int bp = 0x7fffffff;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const unsigned long long key = (pos[q] >= kd) ? lu_key(a[q][j]) : 0ull;
                    const bool better = key > bk;
                    bk = better ? key : bk;
                    bp = better ? pos[q] : bp;
                }
                const unsigned long long wk = lu_wave_max64(bk);
                const int wl = __ffsll((long long)__ballot(bk == wk)) - 1;
#pragma unroll
                for (int q = 0; q < R; ++q)
                    if (lane == wl && pos[q] == bp && wk != 0ull) {      // the WHOLE sub-panel row: the other workgroups need its L part too
#pragma unroll
                        for (int jj = 0; jj < W; ++jj) cdat[buf][wv][jj] = a[q][jj];
                        cand[buf][wv] = LuCand{a[q][j], bp, row0 + tid + LU_NT * q};
                    }
                if (wk == 0ull && lane == 0) cand[buf][wv] = LuCand{0.0, 0x7fffffff, k0};
                __syncthreads();
                if (wv == 0) {                                  // wave 0: this workgroup's winner -> global slot; all G slots -> the panel's winner -> LDS
                    const LuCand cd = cand[buf][lane & 3];
                    const unsigned long long ckey = lu_key(cd.piv);
                    const unsigned int chi = (unsigned int)(ckey >> 32), clo = (unsigned int)ckey;
                    const unsigned int ghi = lu_quad_max32<0x4E>(lu_quad_max32<0xB1>(chi));
                    const unsigned int glo = lu_quad_max32<0x4E>(lu_quad_max32<0xB1>(chi == ghi ? clo : 0u));
                    const unsigned long long lk = ((unsigned long long)ghi << 32) | glo;
                    const int gw = __ffsll((long long)__ballot(ckey == lk)) - 1;
                    {   // this workgroup's record: 18 tagged granules, one per lane
                        unsigned long long bits = 0ull;
                        if (lane < W) bits = (unsigned long long)__double_as_longlong(cdat[buf][gw][lane]);
                        else if (lane == W) bits = (unsigned long long)__double_as_longlong(lk != 0ull ? cand[buf][gw].piv : 0.0);
                        else if (lane == W + 1) bits = (unsigned long long)(unsigned int)cand[buf][gw].pos | ((unsigned long long)(unsigned int)cand[buf][gw].phys << 32);
                        if (lane < W + 2) lu_put(&xb->rec[c0 + j][wg][lane], bits, stamp, near);
                    }
                    {   // all G records (G (W + 2) <= 288 granules): every lane watches up to LU_XK of them, all loads of a look in flight together
                        const int ng = G * (W + 2);
                        const lu_u4* gp[LU_XK];
#pragma unroll
                        for (int u = 0; u < LU_XK; ++u) {
                            const int idx = min(lane + 64 * u, ng - 1), g = idx / (W + 2), k = idx - g * (W + 2);
                            gp[u] = &xb->rec[c0 + j][g][k];
                        }
                        lu_u4 gv[LU_XK];
                        int spins = 0;
                        for (;;) {
                            lu_ld16_n(gp, gv, near);
                            bool ok = true;
#pragma unroll
                            for (int u = 0; u < LU_XK; ++u) ok = ok && (gv[u].z == stamp);
                            if (__ballot(!ok) == 0ull) break;
                            if (++spins > (1 << 20)) { if (lane == 0) atomicOr(status, 4); break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
#pragma unroll
                        for (int u = 0; u < LU_XK; ++u) {
                            const int idx = lane + 64 * u;
                            if (idx < ng) { const int g = idx / (W + 2), k = idx - g * (W + 2); xs[g][k] = ((unsigned long long)gv[u].y << 32) | gv[u].x; }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const double pv = lane < G ? __longlong_as_double((long long)xs[min(lane, G - 1)][W]) : 0.0;
                    const unsigned long long gkey = lu_wave_max64(lu_key(pv));
                    const int win = max(__ffsll((long long)__ballot(lane < G && lu_key(pv) == gkey)) - 1, 0);      // the lowest workgroup among ties
                    if (lane < W) grow[lane] = __longlong_as_double((long long)xs[win][lane]);
                    if (lane == W) {
                        const unsigned long long pp = xs[win][W + 1];
                        LuCand h;
                        h.piv = gkey != 0ull ? __longlong_as_double((long long)xs[win][W]) : 0.0;
                        h.pos = (int)(unsigned int)pp;
                        h.phys = min(max((int)(unsigned int)(pp >> 32), 0), n - 1);       // (never an index out of the matrix, whatever a timed-out poll left)
                        ghdr = h;
                    }
                }
                __syncthreads();
                const LuCand hd = ghdr;
                const int gr = hd.pos;
                if (hd.piv != 0.0) {
                    if (tid == 0) ploc[c0 + j] = hd.phys;
                    double prow[W];
#pragma unroll
                    for (int jj = 0; jj < W; ++jj) prow[jj] = (jj >= j) ? grow[jj] : 0.0;
                    const double rp = lu_rcp(hd.piv);
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        pos[q] = (pos[q] == kd) ? gr : (pos[q] == gr ? kd : pos[q]);
                        const bool act = pos[q] > kd;
                        const double l = act ? a[q][j] * rp : 0.0;
                        a[q][j] = act ? l : a[q][j];
#pragma unroll
                        for (int jj = 0; jj < W; ++jj)
                            if (jj > j) a[q][jj] = fma(-l, prow[jj], a[q][jj]);
                    }
                    if (tid < W) {                               // the pivot row: U part into Ub, the whole sub-panel row into Lp (for (f))
                        if (tid >= j) Ub[c0 + j][c0 + tid] = grow[tid];
                        Lp[j][c0 + tid] = grow[tid];
                    }
                } else {
                    if (tid == 0) { atomicOr(status, 2); ploc[c0 + j] = min(kd, n - 1); }
                    if (tid < W) { Ub[c0 + j][c0 + tid] = 0.0; Lp[j][c0 + tid] = 0.0; }
                }
            }
        });
        // ---- (d): rows go back to their physical places -------------------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (pos[q] >= k0 + c0) {
                double* __restrict__ dst = A + (size_t)(row0 + tid + LU_NT * q) * ld + k0 + c0;
                if (full) lu_st_row<W>(dst, a[q]);
                else {
#pragma unroll
                    for (int jj = 0; jj < W; ++jj) if (c0 + jj < nb) dst[jj] = a[q][jj];
                }
            }
        }
        __syncthreads();                // this workgroup's stores are visible to itself; ploc / Ub / Lp complete
        // ---- (f): owners bring their new pivot rows' right parts up to date and publish them; everybody finishes the U rows -----------
        const int cr0 = c0 + W, ncr = nb - cr0, sp = c0 / W;
        if (ncr > 0) {
            for (int e = tid; e < W * ncr; e += LU_NT) {
                const int i = e / ncr, c = cr0 + (e - i * ncr);
                const int pr = ploc[c0 + i];
                if (pr >= row0 && pr < row0 + R * LU_NT) {       // mine: L entries of earlier sub-panels from my own stores, raw entries from the matrix
                    double t = A[(size_t)pr * ld + k0 + c];
                    for (int kk = 0; kk < c0; ++kk) t = fma(-A[(size_t)pr * ld + k0 + kk], Ub[kk][c], t);
                    lu_put(&xb->trow[sp][i][c], (unsigned long long)__double_as_longlong(t), stamp, near);
                }
            }
            for (int e = tid; e < W * ncr; e += LU_NT) {          // every entry waits for its own granule (its owner may be this workgroup)
                const int i = e / ncr, c = cr0 + (e - i * ncr);
                Tt[i][c] = __longlong_as_double((long long)lu_get(&xb->trow[sp][i][c], stamp, status, near));
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
                    const int pr = ploc[c0 + i];
                    if (pr >= row0 && pr < row0 + R * LU_NT) A[(size_t)pr * ld + k0 + c] = t;
                }
            }
            __syncthreads();
        }
    }
    // ---- the row list: appended by every workgroup (lu_perm_reset cleared the count) ---------------------------------------------------
#pragma unroll
    for (int q = 0; q < R; ++q) {
        if (pos[q] >= 0 && pos[q] != row0 + tid + LU_NT * q) {
            const int e = atomicAdd(&perm->count, 1);
            if (e < LU_MAXTOUCH) { perm->pos[e] = pos[q]; perm->src[e] = row0 + tid + LU_NT * q; }
        }
    }
}

// One workgroup per 64-column slab right of the panel: rows into place, then U12 = L11^-1 A12.
// One round trip: the rows of A12 are read from where they ARE (position k0 + i still sits at row srcof[i]), together with the
// rows that only move; after the barrier the moved rows are written and the triangular solve runs wave-local -- wave w owns
// columns 16 w .. 16 w + 15 of the slab, lane (c16, rq) holds rows rq + 4 u, u < 16, of column c16 in registers, and row i
// reaches the other lanes of its column through one __shfl per step (no barrier, no LDS traffic for B).
__global__ void __launch_bounds__(256) lu_swap_trsm(double* __restrict__ A, int ld, int n, int k0, int nb, const LuPerm* __restrict__ perm, int cend)
{
    __shared__ double Ls[LU_NB][LU_NB + 1];
    __shared__ double Gs[LU_MAXTOUCH][LU_NB];
    __shared__ int spos[LU_MAXTOUCH], ssrc[LU_MAXTOUCH], srcof[LU_NB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c16 = lane & 15, rq = lane >> 4;
    const int cs = k0 + nb + LU_NB * (int)blockIdx.x, wc = min(LU_NB, cend - cs);       // columns [k0 + nb, cend) in slabs of 64
    // (count and rows clamped: after a timed-out hand-off poll of lu_panel_mw -- status bit 4, the solve is then redone or refused -- a list may
    //  hold stale entries; the later kernels of the chain still run on it and must stay inside the matrix)
    const int cnt = perm ? min(perm->count, LU_MAXTOUCH) : 0;
    if (tid < LU_NB) srcof[tid] = k0 + min(tid, nb - 1);
    if (tid < LU_MAXTOUCH) { spos[tid] = tid < cnt ? min(max(perm->pos[tid], k0), n - 1) : 0; ssrc[tid] = tid < cnt ? min(max(perm->src[tid], 0), n - 1) : 0; }
    __syncthreads();
    if (tid < cnt && spos[tid] < k0 + nb) srcof[spos[tid] - k0] = ssrc[tid];
    __syncthreads();
    // rows that only move are staged through LDS, 32 rows (8 loads per thread, unconditional and clamped) per step
    double bcol[16], lv[16];
    const int mycol = cs + min(16 * wv + c16, wc - 1);
#pragma unroll
    for (int u = 0; u < 16; ++u) bcol[u] = A[(size_t)srcof[rq + 4 * u] * ld + mycol];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, r = e >> 6, c = e & 63;
        lv[it] = A[(size_t)(k0 + min(r, nb - 1)) * ld + k0 + min(c, nb - 1)];
    }
    const int gc = tid & 63, gr0 = tid >> 6;
    for (int e0 = 0; e0 < cnt; e0 += 32) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = A[(size_t)ssrc[e0 + gr0 + 4 * u] * ld + cs + min(gc, wc - 1)];      // (entries past cnt read row 0)
#pragma unroll
        for (int u = 0; u < 8; ++u) Gs[e0 + gr0 + 4 * u][gc] = t[u];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, r = e >> 6, c = e & 63;
        Ls[r][c] = (r < nb && c < r) ? lv[it] : 0.0;
    }
    __syncthreads();                                        // every read of the slab is done; Ls and Gs are complete
    for (int e0 = 0; e0 < cnt; e0 += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + gr0 + 4 * u;
            if (e < cnt && gc < wc && spos[e] >= k0 + nb) A[(size_t)spos[e] * ld + cs + gc] = Gs[e][gc];
        }
    }
    lu_static_for<0, LU_NB - 1>([&](auto ic) {      // (a compile-time loop: bcol[] must keep constant indices to stay in registers)
        constexpr int i = decltype(ic)::value, ui = i >> 2, rqi = i & 3;
        const double bi = __shfl(bcol[ui], (rqi << 4) | c16);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u > ui) bcol[u] = fma(-Ls[rq + 4 * u][i], bi, bcol[u]);
            else if (u == ui) bcol[u] = fma(rq > rqi ? -Ls[rq + 4 * u][i] : 0.0, bi, bcol[u]);
        }
        __builtin_amdgcn_sched_barrier(0);              // (without it the scheduler hoists every step's LDS reads to the top: 512 registers and scratch)
    });
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = rq + 4 * u;
        if (r < nb && 16 * wv + c16 < wc) A[(size_t)(k0 + r) * ld + cs + 16 * wv + c16] = bcol[u];
    }
}

// Rows into place on columns [cbeg, cend) (64-column slabs): row perm->pos[e] receives row perm->src[e].  Two-level panels
// (lu.hip) apply the lists of their 16-column groups to the panel's earlier columns and, one list after the other, to everything
// right of the panel.
__global__ void __launch_bounds__(256) lu_apply_perm(double* __restrict__ A, int ld, int n, const LuPerm* __restrict__ perm, int cbeg, int cend)
{
    __shared__ double Gs[LU_MAXTOUCH][LU_NB];
    __shared__ int spos[LU_MAXTOUCH], ssrc[LU_MAXTOUCH];
    const int tid = threadIdx.x, cs = cbeg + LU_NB * (int)blockIdx.x, wc = min(LU_NB, cend - cs);
    const int cnt = min(perm->count, LU_MAXTOUCH);          // (clamped like lu_swap_trsm's: a list behind a timed-out poll may be stale)
    if (tid < LU_MAXTOUCH) { spos[tid] = tid < cnt ? min(max(perm->pos[tid], 0), n - 1) : 0; ssrc[tid] = tid < cnt ? min(max(perm->src[tid], 0), n - 1) : 0; }
    __syncthreads();
    const int gc = tid & 63, gr0 = tid >> 6;
    for (int e0 = 0; e0 < cnt; e0 += 32) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = A[(size_t)ssrc[e0 + gr0 + 4 * u] * ld + cs + min(gc, wc - 1)];      // (entries past cnt read row 0)
#pragma unroll
        for (int u = 0; u < 8; ++u) Gs[e0 + gr0 + 4 * u][gc] = t[u];
    }
    __syncthreads();
    for (int e0 = 0; e0 < cnt; e0 += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + gr0 + 4 * u;
            if (e < cnt && gc < wc) A[(size_t)spos[e] * ld + cs + gc] = Gs[e][gc];
        }
    }
}

// A22 -= L21 U12: tile (blockIdx.y, blockIdx.x) of 64 x 64 below / right of the panel; K = nb <= 64.
#define LU_GS 66
__global__ void __launch_bounds__(256) lu_gemm(double* __restrict__ A, int ld, int n, int k0, int nb, int cend)
{
    __shared__ double Ls[LU_NB][LU_GS];             // [row][k]
    __shared__ double Us[LU_NB][LU_NB + 16];        // [k][column]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, lk = lane >> 4;
    const int wr = wv >> 1, wc = wv & 1;
    const int r0 = k0 + nb + LU_NB * (int)blockIdx.y, c0 = k0 + nb + LU_NB * (int)blockIdx.x;
    const int nr = min(LU_NB, n - r0), nc = min(LU_NB, cend - c0);             // columns [k0 + nb, cend)
    double oldv[2][2][4];                           // the tile of A22 itself: fetched with the operands (one round trip, not two)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 32 * wr + 16 * it + lk + 4 * q, j = 32 * wc + 16 * jt + ln;
                oldv[it][jt][q] = A[(size_t)(r0 + min(i, nr - 1)) * ld + c0 + min(j, nc - 1)];      // (clamped and unconditional: a masked load becomes a branch per element)
            }
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
    // (accumulator element q of lane (ln, lk): row lk + 4 q, column ln -- as chol_syrk)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 32 * wr + 16 * it + lk + 4 * q, j = 32 * wc + 16 * jt + ln;
                if (i < nr && j < nc) A[(size_t)(r0 + i) * ld + c0 + j] = oldv[it][jt][q] - acc[it][jt][q];
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
