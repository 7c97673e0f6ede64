// solver.hpp -- dense solve: blocked Cholesky with border row, back substitution, LU fallback.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_SOLVER_HPP
#define SFFT_AMD_SOLVER_HPP

// ------------------------------------------------------------------------------------------------
// Dense solve.  A is the bordered system [(n+1)][ld] (row n = right hand side), SPD in exact arithmetic
// (it is a Gram matrix).  Right-looking blocked Cholesky on the lower triangle; the
// border row rides along so that the forward substitution L y = b is a by-product (y = row n of L).
// ------------------------------------------------------------------------------------------------
#define CB 64
#define BACK_SLICES 64
// The diagonal block is read from Dsrc ([CB][CB], written by the previous step's trailing update) rather than
// from A, because workgroup 0 overwrites A's diagonal block with the factor while the others may still start.
__global__ void __launch_bounds__(256) chol_copy_diag(const double* __restrict__ A, int ld, int nb, double* __restrict__ Dst)
{
    double v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {            // all 16 loads of a thread first (clamped), then the stores
        const int e = threadIdx.x + 256 * it, i = e >> 6, j = e & 63;
        v[it] = A[(size_t)min(i, nb - 1) * ld + min(j, nb - 1)];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = threadIdx.x + 256 * it, i = e >> 6, j = e & 63;
        if (i < nb && j < nb) Dst[i * CB + j] = v[it];
    }
}

typedef double d4s __attribute__((ext_vector_type(4)));

// A 16 x 16 x 4 fp64 tile product.  (The same product as four v_mfma_f64_4x4x4_4b_f64 with swizzled A operands was measured slower
// in these latency chains -- 0.98 vs 0.90 ms at n = 1735, 7.89 vs 7.70 ms at n = 7207: a lone wave pays more for the eight swizzles
// and four instructions than the shorter pipe occupancy gives back; docs/LOG.md.)
__device__ __forceinline__ d4s mfma16(double av, double bv, d4s acc)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
}

// 1/sqrt(d) to double precision: hardware estimate + two Newton steps (no IEEE division / sqrt sequences on the
// critical path of the factorisation)
// 1 / sqrt(d): v_rsq_f64 is good to about 2^-23 (ISA guide: 2^29 ulp); ONE third-order step -- r (1 + e/2 + 3 e^2/8), e = 1 - d r^2 -- takes that
// below the rounding error in 5 instructions (4 in sequence) where two Newton steps took 7 (6 in sequence): these sit four deep in every
// pivot round of chol_factor_diag, whose time is its instruction count.  d <= 0 or NaN still comes out NaN (rsq: NaN or inf, e: NaN).
__device__ __forceinline__ double rsqrt_nr(double d)
{
    const double r = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * r), r, 1.0);
    return fma(r * e, fma(e, 0.375, 0.5), r);
}

// first node of a solve: a new stamp for this solve's flag hand-offs (never 0 in its upper 24 bits)
// (256 threads: thread t also clears role counter t of chol_panel4, one per outer block)
#define PANEL4_MAX_OUTER 256
__global__ void chol_begin(unsigned int* epoch_ctr, unsigned int* queue, unsigned int* panel_roles)
{
    if (panel_roles && threadIdx.x < PANEL4_MAX_OUTER) panel_roles[threadIdx.x] = 0u;
    if (threadIdx.x != 0) return;
    unsigned int v = (*epoch_ctr + 1u) & 0xFFFFFFu;
    *epoch_ctr = v ? v : 1u;
    *queue = 0u;                            // task counter of chol_dataflow
}

// One panel step.  Every workgroup factors the 64x64 diagonal block (right-looking; thread (i, cg) owns elements
// (i, cg + 4q), q < 16, in registers; four columns per pair of barriers; fully unrolled so that all register
// indices are compile-time and only the triangular part is touched), workgroup 0 stores it, workgroups
// b >= 1 then solve X L^T = A_panel for 64 rows below it (border row n included) without barriers (see below).
// Scratch of the diagonal-block factorisation (one per workgroup)
struct PanelLds {
    double Dl[CB][CB + 1];     // factor of the diagonal block
    double rdiag[CB];          // 1 / L[j][j]
    double Rw[2][CB][4];       // raw column block of a round (ping-pong: round jq writes the block of round jq + 1)
    double Fw[2][CB][4];       // final column block of a round (ping-pong: read by the NEXT round's deferred trailing update)
};

// Factor the CB x CB diagonal block held as a[q] = D[i][cg + 4 q] by thread (i = tid >> 2, cg = tid & 3) (identity padding
// beyond nb); on return L.Dl holds the lower-triangular factor and L.rdiag the reciprocal diagonal.  All 256 threads call.
// Vd (optional, [CB][4]): row 4 jq + k receives row k of the INVERSE of the 4 x 4 pivot block of step jq (used by chol_inv16_mfma).
// MF (chol_dataflow): thread (i = 16 wave + (lane & 15), cg = lane >> 4) instead -- then a[4 t .. 4 t + 3] is the accumulator of a
// v_mfma_f64_16x16x4_f64 whose rows are the columns 16 t .. 16 t + 15 of the block and whose columns are the wave's 16 rows, and the
// rank-4 update of step 4 is one matrix instruction per 16-column tile (A = -F[column][k], zero for the finished columns;
// B = F[row][k], the thread's own final value) instead of up to 60 FMAs and 60 LDS reads; Vd is then computed once after the
// last step from the factor in LDS, not on the way.
template <bool MF = false>
__device__ __forceinline__ void chol_factor_diag(double (&a)[16], PanelLds& L, int tid, int nb, bool report, int* __restrict__ status,
                                                 double (*Vd)[4] = nullptr)
{
    const int i = MF ? 16 * (tid >> 6) + (tid & 15) : tid >> 2, cg = MF ? (tid >> 4) & 3 : tid & 3;
    double (&rdiag)[CB] = L.rdiag;
    // Four columns per round, 16 rounds, ONE barrier per round (round 4: was two -- publish raw / barrier / factor / publish final /
    // barrier / update, 13 us of every 28 us block step of chol_dataflow).  Round jq eliminates columns 4 jq .. 4 jq + 3:
    //   barrier: the raw column block jq of every row (and the final block jq - 1) is visible;
    //   a. deferred trailing update: the finals of round jq - 1 go into the columns right of block jq (off the critical path:
    //      block jq itself received them through the look-ahead of round jq - 1);
    //   b. all threads factor the 4 x 4 pivot block T redundantly (four reciprocal square roots in sequence);
    //   c. thread (i, cg) forward-substitutes its own row through T -> finals L[i][4 jq .. 4 jq + 3] -- and, redundantly, the row
    //      4 (jq + 1) + cg of the NEXT pivot block, which is what its column cg + 4 (jq + 1) needs from this round;
    //   d. look-ahead: its element of column block jq + 1 receives this round's rank-4 update at once and is published as the raw
    //      block of round jq + 1 (ping-pong buffers); the final L[i][4 jq + cg] is published for the next round's step a.
    double pl = 0.0;                                     // own final of the previous round (MF: B operand of the deferred update)
    double pf0 = 0.0, pf1 = 0.0, pf2 = 0.0, pf3 = 0.0;   // own finals of the previous round, all four columns (vector form of the deferred update)
    L.Rw[0][i][cg] = a[0];
#pragma unroll
    for (int jq = 0; jq < 16; ++jq) {
        double (&Rw)[CB][4] = L.Rw[jq & 1];
        double (&Fw)[CB][4] = L.Fw[jq & 1];
        __syncthreads();
        const int j0 = 4 * jq;
        // a. deferred update with the finals of round jq - 1 (buffer (jq - 1) & 1), columns right of block jq
        if (jq >= 1 && jq < 15) {
            double (&Fp)[CB][4] = L.Fw[(jq - 1) & 1];
            if (MF) {
                // every wave updates every tile from the round's first one on -- also the tiles right of its own diagonal tile, whose entries lie
                // above the diagonal and are never read: the barrier waits for wave 3 (which needs them all) anyway, each wave has its own matrix
                // pipe, and without the wave test the round is straight-line code (with it: exec-mask or branch bookkeeping and ~30 register
                // copies per round around the branches)
                const int l15 = tid & 15;
#pragma unroll
                for (int t = (jq + 1) / 4; t < 4; ++t) {
                    const int col = 16 * t + l15;
                    const double fc = Fp[col][cg];                 // (MF: Fw holds -L)
                    const double aop = (col > j0 + 3) ? fc : 0.0;
                    d4s acc = (d4s){a[4 * t], a[4 * t + 1], a[4 * t + 2], a[4 * t + 3]};
                    acc = mfma16(aop, pl, acc);
                    a[4 * t] = acc[0]; a[4 * t + 1] = acc[1]; a[4 * t + 2] = acc[2]; a[4 * t + 3] = acc[3];
                }
            } else {
#pragma unroll
                for (int q = jq + 1; q < 16; ++q) {
                    const int c = cg + 4 * q;
                    a[q] = fma(-pf3, Fp[c][3], fma(-pf2, Fp[c][2], fma(-pf1, Fp[c][1], fma(-pf0, Fp[c][0], a[q]))));
                }
            }
        }
        // b. the pivot block
        const double r00 = Rw[j0][0];
        const double r10 = Rw[j0 + 1][0], r11 = Rw[j0 + 1][1];
        const double r20 = Rw[j0 + 2][0], r21 = Rw[j0 + 2][1], r22 = Rw[j0 + 2][2];
        const double r30 = Rw[j0 + 3][0], r31 = Rw[j0 + 3][1], r32 = Rw[j0 + 3][2], r33 = Rw[j0 + 3][3];
        const double x0 = Rw[i][0], x1 = Rw[i][1], x2 = Rw[i][2], x3 = Rw[i][3];
        const int nr = min(j0 + 4 + cg, CB - 1);         // row of the next pivot block this thread's look-ahead column needs (jq = 15: unused)
        const double y0 = Rw[nr][0], y1 = Rw[nr][1], y2 = Rw[nr][2], y3 = Rw[nr][3];
        const double rs0 = rsqrt_nr(r00);
        const double t10 = r10 * rs0, t20 = r20 * rs0, t30 = r30 * rs0;
        const double d1 = fma(-t10, t10, r11);
        const double rs1 = rsqrt_nr(d1);
        const double t21 = fma(-t20, t10, r21) * rs1, t31 = fma(-t30, t10, r31) * rs1;
        const double d2 = fma(-t21, t21, fma(-t20, t20, r22));
        const double rs2 = rsqrt_nr(d2);
        const double t32 = fma(-t31, t21, fma(-t30, t20, r32)) * rs2;
        const double d3 = fma(-t32, t32, fma(-t31, t31, fma(-t30, t30, r33)));
        const double rs3 = rsqrt_nr(d3);
        if (!MF && tid == 0 && report && j0 < nb && !(r00 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0)) atomicOr(status, 1);      // (MF: checked once, below)
        // c. own row and the look-ahead row through T
        const double l0 = x0 * rs0;
        const double m0 = y0 * rs0;
        const double l1 = fma(-l0, t10, x1) * rs1;
        const double m1 = fma(-m0, t10, y1) * rs1;
        const double l2 = fma(-l1, t21, fma(-l0, t20, x2)) * rs2;
        const double m2 = fma(-m1, t21, fma(-m0, t20, y2)) * rs2;
        const double l3 = fma(-l2, t32, fma(-l1, t31, fma(-l0, t30, x3))) * rs3;
        const double m3 = fma(-m2, t32, fma(-m1, t31, fma(-m0, t30, y3))) * rs3;
        double lf;
        if (MF) {       // cg = lane >> 4 is the DPP row: three row-masked moves instead of three selects on masks that live in spilled scalar registers
            int lo = __double2loint(l0), hi = __double2hiint(l0);
            lo = __builtin_amdgcn_update_dpp(lo, __double2loint(l1), 0xE4, 0x2, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(l1), 0xE4, 0x2, 0xF, false);
            lo = __builtin_amdgcn_update_dpp(lo, __double2loint(l2), 0xE4, 0x4, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(l2), 0xE4, 0x4, 0xF, false);
            lo = __builtin_amdgcn_update_dpp(lo, __double2loint(l3), 0xE4, 0x8, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(l3), 0xE4, 0x8, 0xF, false);
            lf = __hiloint2double(hi, lo);
        } else lf = (cg == 0) ? l0 : (cg == 1) ? l1 : (cg == 2) ? l2 : l3;
        // d. look-ahead: column cg + 4 (jq + 1) of this row, then publish it as the next round's raw block
        if (jq < 15) {
            a[jq + 1] = fma(-l3, m3, fma(-l2, m2, fma(-l1, m1, fma(-l0, m0, a[jq + 1]))));
            L.Rw[(jq + 1) & 1][i][cg] = a[jq + 1];
        }
        a[jq] = lf;                                      // final L[i][4 jq + cg] (entries above the diagonal: unused garbage)
        Fw[i][cg] = MF ? -lf : lf;
        pl = lf; pf0 = l0; pf1 = l1; pf2 = l2; pf3 = l3;
        if (MF) { if (tid == 0) { rdiag[j0] = rs0; rdiag[j0 + 1] = rs1; rdiag[j0 + 2] = rs2; rdiag[j0 + 3] = rs3; } }      // (every thread has all four)
        else if (tid < 4) rdiag[j0 + tid] = (tid == 0) ? rs0 : (tid == 1) ? rs1 : (tid == 2) ? rs2 : rs3;
        if (!MF && Vd && (tid >> 4) == 4) {         // sixteen lanes of wave 1 (off the stores above): element (k, c) of the inverse of the pivot block
            // [[1/rs0], [t10, 1/rs1], [t20, t21, 1/rs2], [t30, t31, t32, 1/rs3]]
            const int k = (tid >> 2) & 3, c = tid & 3;
            const double v10 = -t10 * rs0 * rs1, v21 = -t21 * rs1 * rs2, v32 = -t32 * rs2 * rs3;
            const double v20 = -fma(t21, v10, t20 * rs0) * rs2, v31 = -fma(t32, v21, t31 * rs1) * rs3;
            const double v30 = -fma(t32, v20, fma(t31, v10, t30 * rs0)) * rs3;
            const double r0v = (c == 0) ? rs0 : 0.0;
            const double r1v = (c == 0) ? v10 : (c == 1) ? rs1 : 0.0;
            const double r2v = (c == 0) ? v20 : (c == 1) ? v21 : (c == 2) ? rs2 : 0.0;
            const double r3v = (c == 0) ? v30 : (c == 1) ? v31 : (c == 2) ? v32 : rs3;
            Vd[j0 + k][c] = (k == 0) ? r0v : (k == 1) ? r1v : (k == 2) ? r2v : r3v;
        }
    }
    __syncthreads();                                     // (rdiag of the last round)
    if (MF && report && tid < nb) {
        // a pivot that is not positive (or not a number) leaves NaN in its reciprocal square root -- v_rsq_f64 of d <= 0 is NaN or infinite and
        // the Newton steps turn both into NaN -- so one look at rdiag replaces four compares per round
        const double rsd = rdiag[tid];
        if (!(rsd > 0.0 && rsd < __builtin_huge_val())) atomicOr(status, 1);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = cg + 4 * q;
        L.Dl[i][c] = (c <= i) ? a[q] : 0.0;
    }
    __syncthreads();
    if (MF && Vd) {
        // element (k, c) of the inverse of the 4 x 4 pivot block b = [[1/rs0], [t10, 1/rs1], [t20, t21, 1/rs2], [t30, t31, t32, 1/rs3]]
        const int b = tid >> 4, k = (tid >> 2) & 3, c = tid & 3, j0 = 4 * b;
        const double rs0 = rdiag[j0], rs1 = rdiag[j0 + 1], rs2 = rdiag[j0 + 2], rs3 = rdiag[j0 + 3];
        const double t10 = L.Dl[j0 + 1][j0], t20 = L.Dl[j0 + 2][j0], t21 = L.Dl[j0 + 2][j0 + 1];
        const double t30 = L.Dl[j0 + 3][j0], t31 = L.Dl[j0 + 3][j0 + 1], t32 = L.Dl[j0 + 3][j0 + 2];
        const double v10 = -t10 * rs0 * rs1, v21 = -t21 * rs1 * rs2, v32 = -t32 * rs2 * rs3;
        const double v20 = -fma(t21, v10, t20 * rs0) * rs2, v31 = -fma(t32, v21, t31 * rs1) * rs3;
        const double v30 = -fma(t32, v20, fma(t31, v10, t30 * rs0)) * rs3;
        const double r0v = (c == 0) ? rs0 : 0.0;
        const double r1v = (c == 0) ? v10 : (c == 1) ? rs1 : 0.0;
        const double r2v = (c == 0) ? v20 : (c == 1) ? v21 : (c == 2) ? rs2 : 0.0;
        const double r3v = (c == 0) ? v30 : (c == 1) ? v31 : (c == 2) ? v32 : rs3;
        Vd[j0 + k][c] = (k == 0) ? r0v : (k == 1) ? r1v : (k == 2) ? r2v : r3v;
        __syncthreads();
    }
}

// X L^T = P for the workgroup's CB rows, thread (i, cg) holding P[i][cg + 4 q] in p[q]; no barriers
__device__ __forceinline__ void chol_trsm_rows(double (&p)[16], const PanelLds& L, int tid)
{
    const int cg = tid & 3;
    const double (&Dl)[CB][CB + 1] = L.Dl;
    const double (&rdiag)[CB] = L.rdiag;
    // X L^T = A_panel, row by row: x_j = (a_j - sum_{t<j} x_t L[j][t]) / L[j][j].  The four lanes of a row each hold
    // the x_t with t = cg (mod 4); they form partial sums over their own t and combine them with two quad
    // shuffles, so this phase needs no barrier at all (rows are independent).
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int jq = j >> 2, jr = j & 3;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int q = 0; q < jq; ++q) {
            const double lv = Dl[j][cg + 4 * q];
            if (q & 1) acc1 = fma(p[q], lv, acc1); else acc0 = fma(p[q], lv, acc0);
        }
        {   // columns 4 jq + cg < j only
            const double lv = (cg < jr) ? Dl[j][cg + 4 * jq] : 0.0;
            acc0 = fma(p[jq], lv, acc0);
        }
        double tot = acc0 + acc1;
        tot += __shfl_xor(tot, 1);
        tot += __shfl_xor(tot, 2);
        const double xj = (p[jq] - tot) * rdiag[j];
        p[jq] = (cg == jr) ? xj : p[jq];
    }
}

__global__ void __launch_bounds__(256) chol_panel(double* __restrict__ A, int ld, int n, int k, const double* __restrict__ Dsrc,
                                                  int* __restrict__ status, double* __restrict__ rd)
{
    __shared__ PanelLds L;
    const int tid = threadIdx.x;
    const int nb = min(CB, n - k);
    const int i = tid >> 2, cg = tid & 3;
    double a[16];
    {   // (loads first, a scheduling fence, then the masks: see chol_inv_diag)
        double dv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) dv[q] = Dsrc[i * CB + cg + 4 * q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = cg + 4 * q;
            a[q] = (i < nb && c <= i) ? dv[q] : ((i == c) ? 1.0 : 0.0);           // identity padding beyond nb
        }
    }
    // this workgroup's panel rows are requested now, so that their latency hides behind the factorization of the diagonal
    // block (the barriers below would otherwise keep the loads after it)
    const int r0 = k + nb + ((int)blockIdx.x - 1) * CB;
    const int nr = blockIdx.x == 0 ? 0 : min(CB, n + 1 - r0);
    double p[16];
    {
        double pv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) pv[q] = A[(size_t)max(min(r0 + i, n), 0) * ld + k + min(cg + 4 * q, nb - 1)];      // clamped (workgroup 0 of a partial block: r0 < 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 16; ++q) p[q] = (i < nr && cg + 4 * q < nb) ? pv[q] : 0.0;
    }
    chol_factor_diag(a, L, tid, nb, blockIdx.x == 0, status);
    if (blockIdx.x == 0) {
        for (int e = tid; e < nb * nb; e += 256) {
            const int r = e / nb, c = e - r * nb;
            if (c <= r) A[(size_t)(k + r) * ld + k + c] = L.Dl[r][c];
        }
        if (tid < nb) rd[k + tid] = L.rdiag[tid];
        return;
    }
    if (nr <= 0) return;
    chol_trsm_rows(p, L, tid);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = cg + 4 * q;
        if (i < nr && c < nb) A[(size_t)(r0 + i) * ld + k + c] = p[q];
    }
}

__global__ void __launch_bounds__(256) chol_update(double* __restrict__ A, int ld, int n, int k, double* __restrict__ Dnext)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    __shared__ double Li[CB][CB + 1];
    __shared__ double Lj[CB][CB + 1];
    const int tid = threadIdx.x;
    const int nb = min(CB, n - k);
    const int r0 = k + nb;
    const int i0 = r0 + ti * CB, j0 = r0 + tj * CB;
    const int ni = min(CB, n + 1 - i0), nj = min(CB, n - j0);
    if (ni <= 0 || nj <= 0) return;
#pragma unroll
    for (int it = 0; it < 16; ++it) {                 // 32 independent loads in flight per thread
        const int e = tid + 256 * it;
        const int i = e >> 6, t = e & 63;
        const double vi = A[(size_t)(i0 + min(i, ni - 1)) * ld + k + min(t, nb - 1)];      // (clamped, then masked)
        const double vj = A[(size_t)(j0 + min(i, nj - 1)) * ld + k + min(t, nb - 1)];
        Li[i][t] = (i < ni && t < nb) ? vi : 0.0;
        Lj[i][t] = (i < nj && t < nb) ? vj : 0.0;
    }
    const int tx = tid & 15, ty = tid >> 4;
    // the entries to be updated are requested before the barrier and the product loop, which then hide their latency
    double old[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ty + 16 * r, j = tx + 16 * q;
            const bool ok = (i < ni) && (j < nj) && (j0 + j <= i0 + i);
            const double ov = A[(size_t)(i0 + min(i, ni - 1)) * ld + j0 + min(j, nj - 1)];      // (clamped, then masked)
            old[r][q] = ok ? ov : 0.0;
        }
    __syncthreads();
    double c[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) c[r][q] = 0.0;
#pragma unroll 8
    for (int t = 0; t < CB; ++t) {                    // columns t >= nb are zero padded
        double av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = Li[ty + 16 * r][t];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = Lj[tx + 16 * q][t];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) c[r][q] = fma(av[r], bv[q], c[r][q]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ty + 16 * r, j = tx + 16 * q;
            const bool ok = (i < ni) && (j < nj) && (j0 + j <= i0 + i);   // lower triangle only
            if (ok) {
                const double v = old[r][q] - c[r][q];
                A[(size_t)(i0 + i) * ld + j0 + j] = v;
                if (ti == 0 && tj == 0) Dnext[i * CB + j] = v;          // next step's diagonal block
            }
        }
}

// Back substitution L^T x = y (y = border row n of the factor), one launch per 64-row block, last block first.
// x_b = L_bb^-T ( y_b - sum_{rows below} L[row][b]^T x[row] ).  The strip product is spread over gridDim.x
// workgroups (64 rows each); the last one to arrive (device-scope counter) reduces the partials and solves the
// 64x64 triangle.  xv is [n] (stripe-free ordering).
__global__ void __launch_bounds__(256) chol_back_step(const double* __restrict__ A, int ld, int n, int kb, double* __restrict__ xv,
                                                      double* __restrict__ partial, unsigned int* __restrict__ counter,
                                                      const double* __restrict__ rd)
{
    __shared__ double red[4][CB];
    __shared__ double D[CB][CB + 1];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int nb = min(CB, n - kb);
    const int c = tid & 63, rg = tid >> 6;
    const int rows_below = n - (kb + nb);
    const int nslice = gridDim.x;
    if (rows_below > 0) {
        const int per = (rows_below + nslice - 1) / nslice;
        const int rb = kb + nb + blockIdx.x * per;
        const int re = min(n, rb + per);
        double acc = 0.0;
        if (c < nb) {
#pragma unroll 8
            for (int row = rb + rg; row < re; row += 4) acc = fma(A[(size_t)row * ld + kb + c], xv[row], acc);
        }
        red[rg][c] = acc;
        __syncthreads();
        if (tid < CB) partial[(size_t)blockIdx.x * CB + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        __syncthreads();
        if (tid == 0) {                                   // one lane releases for the workgroup
            __threadfence();
            const unsigned int prev = atomicAdd(counter, 1u);
            is_last = (prev == (unsigned int)(nslice - 1));
            if (is_last) __threadfence();                 // ... and acquires for the last arriver
        }
        __syncthreads();
        if (!is_last) return;
    } else if (blockIdx.x != 0) return;
    // last arriver: y_b - strip product, then the triangle
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it;
        const int r = e >> 6, q = e & 63;
        D[r][q] = (r < nb && q < nb) ? A[(size_t)(kb + r) * ld + kb + q] : 0.0;
    }
    double ps = 0.0;
    if (rows_below > 0 && c < nb)
        for (int g = rg; g < nslice; g += 4) ps += __hip_atomic_load(&partial[(size_t)g * CB + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[rg][c] = ps;
    __syncthreads();
    if (tid < 64) {
        double yt = 0.0, rdj = 1.0;
        if (tid < nb) {
            yt = A[(size_t)n * ld + kb + tid] - (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
            rdj = rd[kb + tid];
        }
        for (int j = nb - 1; j >= 0; --j) {
            const double xj = __shfl(yt, j) * __shfl(rdj, j);
            if (tid == j) yt = xj;
            else if (tid < j) yt = fma(-D[j][tid], xj, yt);
        }
        if (tid < nb) xv[kb + tid] = yt;
        if (tid == 0) *counter = 0u;
    }
}

// Extend_Solution / Restore_Solution scatter (SFFTConfigure.py:1299-1311; BSplineSFFT.py:2274-2338):
// solution[idx[i]] = x[i]; removed entries stay zero, tied entries all receive the value of their representative
// ---- one launch per block step -----------------------------------------------------------------------------------
// chol_step(k): trailing update with the panel of step k - CB (as chol_update), and, in the tiles of the first trailing block
// column, the panel work of step k on the freshly updated values (as chol_panel): tile (0, 0) publishes the updated, not yet
// factored diagonal block (global scratch + an epoch-stamped flag; it is the first workgroup of the grid), every tile of that
// block column then factors it redundantly and solves its own 64 rows.  Halves the number of dependent launches of the
// factorisation (update and panel used to be 20 + 27 us each, most of it launch / first-load latency).  Only for full blocks
// (n - k >= CB); the last, partial block keeps the two-kernel path.

__global__ void __launch_bounds__(256) chol_step(double* __restrict__ A, int ld, int n, int kp, double* Draw, unsigned int* flag,
                                                 const unsigned int* __restrict__ epoch_ctr, unsigned int step_id, int* __restrict__ status,
                                                 double* __restrict__ rd)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    // stamp of this hand-off: (solve counter, step) -- the counter lives in device memory and is bumped by chol_begin at the
    // start of every solve, so that the whole chain of launches has constant arguments and replays as one hipGraph
    const unsigned int epoch = (*epoch_ctr << 8) | step_id;
    __shared__ double smem[2 * CB * (CB + 1) + CB + 16 * CB + 16];
    double (*Li)[CB + 1] = reinterpret_cast<double (*)[CB + 1]>(smem);                       // later: the updated tile T
    double (*Lj)[CB + 1] = reinterpret_cast<double (*)[CB + 1]>(smem + CB * (CB + 1));       // later: start of the PanelLds
    PanelLds& L = *reinterpret_cast<PanelLds*>(smem + CB * (CB + 1));
    const int tid = threadIdx.x;
    const int k = kp + CB;                  // this step's first column = start of the trailing matrix
    const int i0 = k + ti * CB, j0 = k + tj * CB;
    const int ni = min(CB, n + 1 - i0), nj = min(CB, n - j0);
    if (ni <= 0 || nj <= 0) return;
    // (unconditional loads from clamped rows, masked afterwards: a `cond ? A[..] : 0` load is compiled as a branch around the
    //  load with a full s_waitcnt behind it -- 32 dependent round trips per thread, most of the 39 us this kernel used to take)
    {
        double vi[16], vj[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it;
            const int i = e >> 6, t = e & 63;
            vi[it] = A[(size_t)(i0 + min(i, ni - 1)) * ld + kp + t];
            vj[it] = A[(size_t)(j0 + min(i, nj - 1)) * ld + kp + t];
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it;
            const int i = e >> 6, t = e & 63;
            Li[i][t] = (i < ni) ? vi[it] : 0.0;
            Lj[i][t] = (i < nj) ? vj[it] : 0.0;
        }
    }
    // rank-CB update on the matrix cores: wave wv owns rows 16 wv .. 16 wv + 15 of the tile, four 16 x 16 column tiles, 16
    // k-steps of v_mfma_f64_16x16x4_f64 each (A[i][k] = Li[16 wv + i][4 ks + k], B[k][j] = Lj[16 jt + j][4 ks + k]).
    // Thread holds C[16 wv + (lane >> 4) + 4 q][16 jt + (lane & 15)] in c[jt][q].
    const int lane = tid & 63, wv = tid >> 6, ln = lane & 15, lk = lane >> 4;
    double old[4][4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 16 * wv + lk + 4 * q, j = 16 * jt + ln;
            const bool ok = (i < ni) && (j < nj) && (j0 + j <= i0 + i);
            const double v = A[(size_t)(i0 + min(i, ni - 1)) * ld + j0 + min(j, nj - 1)];      // (clamped, then masked: see above)
            old[jt][q] = ok ? v : 0.0;
        }
    __syncthreads();
    d4s c[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) c[jt] = (d4s){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int ks = 0; ks < CB / 4; ++ks) {
        const double av = Li[16 * wv + ln][4 * ks + lk];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) c[jt] = mfma16(av, Lj[16 * jt + ln][4 * ks + lk], c[jt]);
    }
    if (tj != 0) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * wv + lk + 4 * q, j = 16 * jt + ln;
                if ((i < ni) && (j < nj) && (j0 + j <= i0 + i)) A[(size_t)(i0 + i) * ld + j0 + j] = old[jt][q] - c[jt][q];
            }
        return;
    }
    // ---- first trailing block column: the panel of step k (nb = nj = CB columns) --------------------------------------
    __syncthreads();                        // everyone is done reading Li / Lj
    double (*T)[CB + 1] = Li;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) T[16 * wv + lk + 4 * q][16 * jt + ln] = old[jt][q] - c[jt][q];      // (entries outside the tile: 0 - 0)
    __syncthreads();
    const int i = tid >> 2, cg = tid & 3;
    double a[16], p[16];
    if (ti == 0) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it;
            Draw[e] = T[e >> 6][e & 63];
        }
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cc = cg + 4 * q;
            a[q] = (cc <= i) ? T[i][cc] : ((i == cc) ? 1.0 : 0.0);
            p[q] = 0.0;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) p[q] = T[i][cg + 4 * q];
        if (tid == 0) {      // bounded spin (~ seconds): a protocol error must not hang the device; it reports through `status` -> LU fallback
            long long spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1LL << 26)) { atomicOr(status, 4); break; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cc = cg + 4 * q;
            a[q] = (cc <= i) ? __hip_atomic_load(&Draw[i * CB + cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((i == cc) ? 1.0 : 0.0);
        }
    }
    __syncthreads();                        // T has been read; the PanelLds region (old Lj) may be overwritten
    chol_factor_diag(a, L, tid, CB, ti == 0, status);
    if (ti == 0) {
        for (int e = tid; e < CB * CB; e += 256) {
            const int r = e >> 6, cc = e & 63;
            if (cc <= r) A[(size_t)(k + r) * ld + k + cc] = L.Dl[r][cc];
        }
        if (tid < CB) rd[k + tid] = L.rdiag[tid];
        return;
    }
    chol_trsm_rows(p, L, tid);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int cc = cg + 4 * q;
        if (i < ni) A[(size_t)(i0 + i) * ld + k + cc] = p[q];
    }
}

// ---- the whole factorisation as ONE launch: dataflow over 64 x 64 tiles ------------------------------------------------
// Left-looking tile Cholesky (with the border row riding along) executed by a fixed number of persistent workgroups that
// pull tasks from a device-side queue (an atomic counter) in a topological order; a task waits for the tiles it consumes
// through one epoch-stamped flag per tile (release by the producer after its stores, acquire by the consumer's lane 0,
// agent scope) -- no host round trips, no launch boundaries, and the panel of step j + 1 overlaps the updates of step j.
//
//   PT(j):  owns the sub-diagonal tile X = (j, j-1) and the diagonal tile D = (j, j).  It accumulates
//           X -= L(j,k) L(j-1,k)^T and D -= L(j,k) L(j,k)^T for k < j-1 while those tiles arrive, then -- the critical path --
//           waits for the factor of D(j-1), solves X L(j-1,j-1)^T = X, applies D -= X X^T from its own registers, factors D
//           and publishes both.  One hand-off per block step instead of two (trsm -> update -> factor stay in one workgroup).
//   TR(i,j): every other tile (rows i >= j+2 and the border row): C = A(i,j) - sum_{k<j} L(i,k) L(j,k)^T, then X L(j,j)^T = C.
//
// Task order: PT(0); then for j = 1 .. nbc-1: PT(j), TR(j+1 .. nbc-1, j-1), TR(border, j-1); last TR(border, nbc-1).  Every
// task only waits for tasks earlier in the order, and a task is only ever taken by a workgroup that is already running, so
// the kernel cannot deadlock however few of its workgroups are resident (other streams' kernels may hold the other CUs);
// all spins are bounded and report through `status` (-> pivoted-LU fallback) rather than hang.
// Tile rows are whole 128-byte lines (ld is a multiple of 16 doubles, tiles start at multiples of 64 columns), so a tile's
// final values never share a line with data another workgroup is still rewriting.
#define DF_EPOCH_TAG 0x7Eu
struct DfGeom { int n, nbc, ld; };
__device__ __forceinline__ int df_flag_id(const DfGeom& g, int i, int j) { return j * (g.nbc + 1) + i; }    // i = nbc: the border row

// Hand-off protocol (placement independent; the per-XCD L2s are not coherent and a CU's L1 is never refreshed by other CUs):
//   producer: payload with write-through (sc1) 16-byte stores, every storing wave drains its stores (s_waitcnt vmcnt(0)),
//             __syncthreads(), then ONE lane stores the flag (relaxed, agent scope) -- no release fence needed;
//   consumer: ONE lane polls the flag with relaxed agent-scope loads (+ s_sleep), ONE agent-scope acquire after the match
//             (drops this CU's stale L1 lines), __syncthreads(), then plain loads by every wave.
typedef unsigned int df_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void df_poll(const unsigned int* flags, int id, unsigned int epoch, int* __restrict__ status)
{
    long long spins = 0;
    while (__hip_atomic_load(flags + id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1LL << 26)) { atomicOr(status, 4); break; }      // bounded: a protocol error reports (-> LU fallback), never hangs
    }
}
__device__ __forceinline__ void df_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

// all threads call: drain this wave's payload stores, meet, then lane 0 stamps the flag
__device__ __forceinline__ void df_publish(unsigned int* flags, int id, unsigned int epoch, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + id, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// write-through store of the LDS tile T (rows < nr) to the tile of A at (r0, c0): 16 bytes per lane, whole rows of 512 bytes
// (columns beyond the system's last one land in the row padding -- never past the row -- or in the unused upper triangle)
__device__ __forceinline__ void df_store_tile(__amdgpu_buffer_rsrc_t rsrc, int ld, int r0, int nr, int c0, const double (*T)[CB + 1], int tid)
{
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = tid + 256 * it, r = e >> 5, c = 2 * (e & 31);
        const double v0 = T[r][c], v1 = T[r][c + 1];
        df_u4 pk;
        pk.x = (unsigned int)__double2loint(v0); pk.y = (unsigned int)__double2hiint(v0);
        pk.z = (unsigned int)__double2loint(v1); pk.w = (unsigned int)__double2hiint(v1);
        if (r < nr && c0 + c < ld) __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, (int)((((size_t)(r0 + r)) * ld + c0 + c) * sizeof(double)), 0, /*aux: sc1*/ 16);
    }
}

// load the 64 x 64 tile with first row r0 (nr valid rows, the rest read as zero) and first column c0 into LDS [64][65]
__device__ __forceinline__ void df_load_tile(const double* __restrict__ A, int ld, int r0, int nr, int c0, double (*dst)[CB + 1], int tid)
{
    double v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, i = e >> 6, t = e & 63;
        v[it] = A[(size_t)(r0 + min(i, nr - 1)) * ld + c0 + t];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, i = e >> 6, t = e & 63;
        dst[i][t] = (i < nr) ? v[it] : 0.0;
    }
}


// Inverses of the four 16 x 16 diagonal sub-blocks of the factor held in L.Dl (with L.rdiag), by ONE wave: lane 16 b + c owns
// column c of W_bb = L_bb^-1 and runs the forward substitution down that column (16 steps, all lanes in lockstep).  Result in
// W16t[b][c][i] = W_bb[i][c].  These are what lets every triangular solve against this block run on the matrix cores.
#define W16_LD 17
__device__ __forceinline__ void chol_inv16(const PanelLds& L, double (*W16t)[16][W16_LD], int lane)
{
    const int b = lane >> 4, c = lane & 15;
    double w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s0 = (i == c) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
        for (int t = 0; t < i; ++t) {
            const double lv = L.Dl[16 * b + i][16 * b + t];
            if (t & 1) s1 = fma(-lv, w[t], s1); else s0 = fma(-lv, w[t], s0);
        }
        w[i] = (i >= c) ? (s0 + s1) * L.rdiag[16 * b + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) W16t[b][c][i] = w[i];
}

// The same four inverses on the matrix cores, one 16 x 16 block per wave (b = wave index): with D = blockdiag of the block's
// 4 x 4 diagonal sub-blocks (inverses in Vd, from chol_factor_diag) L_bb = D (I + N), N = D^-1 (L_bb - D) strictly block-lower,
// hence nilpotent: N^4 = 0 and L_bb^-1 = (I - N + N^2 - N^3) D^-1.  Four 16^3 products = 16 v_mfma_f64_16x16x4_f64 and three
// trips through the wave's LDS scratch Ws[3][16][17]; the result lands in W16t[b][c][i] = W_bb[i][c].
__device__ __forceinline__ void chol_inv16_mfma(const double (*Dl)[CB + 1], const double (*Vd)[4], double (*Ws)[16][W16_LD],
                                                double (*W16t)[16][W16_LD], int b, int ln, int lk)
{
    const int R = 16 * b;
    d4s nA = (d4s){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const double av = ((ln >> 2) == ks) ? Vd[R + ln][lk] : 0.0;                          // D^-1 [i = ln][k = 4 ks + lk]
        const double bv = (ks > (ln >> 2)) ? Dl[R + 4 * ks + lk][R + ln] : 0.0;              // (L_bb - D)[k][j = ln]
        nA = mfma16(av, bv, nA);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Ws[0][lk + 4 * q][ln] = nA[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    d4s n2 = (d4s){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) n2 = mfma16(Ws[0][ln][4 * ks + lk], Ws[0][4 * ks + lk][ln], n2);
#pragma unroll
    for (int q = 0; q < 4; ++q) Ws[1][lk + 4 * q][ln] = n2[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    d4s n3 = (d4s){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) n3 = mfma16(Ws[0][ln][4 * ks + lk], Ws[1][4 * ks + lk][ln], n3);
#pragma unroll
    for (int q = 0; q < 4; ++q) Ws[2][lk + 4 * q][ln] = ((lk + 4 * q == ln) ? 1.0 : 0.0) - nA[q] + n2[q] - n3[q];      // I - N + N^2 - N^3
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    d4s wacc = (d4s){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const double bv = (ks == (ln >> 2)) ? Vd[R + 4 * ks + lk][ln & 3] : 0.0;             // D^-1 [k][j = ln]
        wacc = mfma16(Ws[2][ln][4 * ks + lk], bv, wacc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) W16t[b][ln][lk + 4 * q] = wacc[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// W = L^-1 (64 x 64, lower triangular) from L (Dl) and the inverses of its four 16 x 16 diagonal sub-blocks (W16t), by block forward
// substitution on the matrix cores: column-block c is owned by wave c (c < 3), W_rc = -W_rr sum_{m = c}^{r-1} L_rm W_mc for r = c+1 .. 3.
// Result in Wf[64][65] (zero above the diagonal).  What chol_inv_diag computes in a launch of its own (64 barrier-separated steps);
// here it runs in the factoring workgroup AFTER the factor has been published, i.e. off the critical path of the factorisation.
// Sm: per-wave scratch [4][16][17].  All 256 threads call (barriers inside).
__device__ __forceinline__ void chol_inv64_mfma(const double (*Dl)[CB + 1], const double (*W16t)[16][W16_LD], double (*Wf)[CB + 1],
                                                double (*Sm)[16][W16_LD], int tid, int wv, int ln, int lk)
{
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e >> 6, c = e & 63;
        Wf[r][c] = ((r >> 4) == (c >> 4)) ? W16t[r >> 4][c & 15][r & 15] : 0.0;
    }
    __syncthreads();
    if (wv < 3) {
        const int c = wv;
        for (int r = c + 1; r < 4; ++r) {
            d4s acc = (d4s){0.0, 0.0, 0.0, 0.0};
            for (int m = c; m < r; ++m)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc = mfma16(Dl[16 * r + ln][16 * m + 4 * ks + lk], Wf[16 * m + 4 * ks + lk][16 * c + ln], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) Sm[wv][lk + 4 * q][ln] = acc[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            d4s x = (d4s){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                x = mfma16(W16t[r][4 * ks + lk][ln], Sm[wv][4 * ks + lk][ln], x);      // A[i][k] = W_rr[i][k]
#pragma unroll
            for (int q = 0; q < 4; ++q) Wf[16 * r + lk + 4 * q][16 * c + ln] = -x[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
}

// X L^T = C for the 64 x 64 tile C held in T (in place), L in Dl (identity beyond its last row), W16t as above.  Each wave owns
// 16 rows of T and needs no workgroup barrier: for the four 16-column blocks in turn, Y_b = C_b - sum_{b' < b} X_b' L_bb'^T and
// X_b = Y_b W_bb^T, all as v_mfma_f64_16x16x4_f64 (40 per wave).  The 16 x 16 inverses are only used block-diagonally; the
// coupling between blocks is by substitution.
__device__ __forceinline__ void df_trsm_mfma(double (*T)[CB + 1], const double (*Dl)[CB + 1], const double (*W16t)[16][W16_LD], int wv, int ln, int lk)
{
    const int R0 = 16 * wv;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        d4s acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = T[R0 + lk + 4 * q][16 * b + ln];
#pragma unroll
        for (int bp = 0; bp < b; ++bp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = mfma16(-T[R0 + ln][16 * bp + 4 * ks + lk], Dl[16 * b + ln][16 * bp + 4 * ks + lk], acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) T[R0 + lk + 4 * q][16 * b + ln] = acc[q];           // Y_b
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        d4s x = (d4s){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            x = mfma16(T[R0 + ln][16 * b + 4 * ks + lk], W16t[b][4 * ks + lk][ln], x);     // B[k][j] = W_bb[j][k]
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) T[R0 + lk + 4 * q][16 * b + ln] = x[q];             // X_b
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void __launch_bounds__(256) chol_dataflow(double* __restrict__ A, int ld, int n, unsigned int* flags, unsigned int* queue,
                                                     const unsigned int* __restrict__ epoch_ctr, int* __restrict__ status,
                                                     double* __restrict__ rd, double* __restrict__ w16, double* __restrict__ winv, unsigned long long* __restrict__ trace)
{
    __shared__ double smem[2 * CB * (CB + 1) + CB + 16 * CB + 16 + 4 * 16 * W16_LD + 4 * CB];
    __shared__ int s_task;
#define DF_TRACE(slot) do { if (trace && tid == 0) trace[(size_t)bj * 16 + (slot)] = wall_clock64(); } while (0)
    double (*Li)[CB + 1] = reinterpret_cast<double (*)[CB + 1]>(smem);                       // operand tile / the accumulated tile T
    double (*Lj)[CB + 1] = reinterpret_cast<double (*)[CB + 1]>(smem + CB * (CB + 1));       // operand tile / start of the PanelLds
    PanelLds& L = *reinterpret_cast<PanelLds*>(smem + CB * (CB + 1));
    double (*W16t)[16][W16_LD] = reinterpret_cast<double (*)[16][W16_LD]>(smem + 2 * CB * (CB + 1) + CB + 16 * CB + 16);
    double (*Vd)[4] = reinterpret_cast<double (*)[4]>(smem + 2 * CB * (CB + 1) + CB + 16 * CB + 16 + 4 * 16 * W16_LD);
    // chol_inv64_mfma's per-wave scratch (waves 0..2 only: 3 x 16 x 17 doubles) lies over the panel's Rw/Fw (2 x 512 doubles), which are
    // dead once chol_factor_diag has returned; keeping it out of the static size leaves room for a row-pass workgroup beside this one on a CU
    static_assert(3 * 16 * W16_LD <= 2 * 2 * CB * 4, "Sm64 overlay does not fit into PanelLds::Rw + Fw");
    double (*Sm64)[16][W16_LD] = reinterpret_cast<double (*)[16][W16_LD]>(&L.Rw[0][0][0]);
    const unsigned int epoch = (*epoch_ctr << 8) | DF_EPOCH_TAG;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, lk = lane >> 4;
    const int ti = 16 * wv + ln, cg = lk;                    // element ownership of chol_factor_diag<true>
    DfGeom g;
    g.n = n; g.ld = ld; g.nbc = (n + CB - 1) / CB;
    const int nbc = g.nbc;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((size_t)(n + 1) * ld * sizeof(double)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)w16, 0, (int)((size_t)nbc * 1024 * sizeof(double)), 0x00020000);
    // The diagonal chain PT(0), PT(1), ... is the critical path.  It has DEDICATED workgroups -- the first npt of the grid, PT(j) on
    // workgroup j mod npt -- so that PT(j) starts accumulating its look-ahead products npt - 1 steps before it is due, however busy
    // the pool is.  (Round 3 handed PT tasks out through the same queue as the TR tasks: the per-step stamps of SFFT_DF_TRACE showed
    // 8 of 28 steps at n = 1735 starting 6 - 20 us late because their PT task had been picked up too late -- 690 us for 28 steps of
    // 21 us.)  A PT task's accumulation is (j - 1) x 128 matrix instructions, ~7 us per k: npt = 2 + nbc / 3 gives every PT(j) the
    // lead it needs (3 workers: 1264 us, the accumulation becomes the chain).  Every other workgroup pulls TR tasks from the queue; a PT
    // workgroup joins them when its diagonal tasks are done.  Progress: block indices are dispatched in order, so the PT workgroups
    // are resident before any TR workgroup; a TR task only waits for PT tasks and for TR tasks earlier in the queue, all of which have
    // been taken by running workgroups.
    if (trace && tid == 0 && blockIdx.x == 0) trace[11] = wall_clock64();                       // (row 0, slot 11: kernel start)
    if (gridDim.x < 2) {         // the PT / TR split needs a queue workgroup beside the diagonal one: report (-> LU), never spin
        if (tid == 0) atomicOr(status, 8);
        return;
    }
    const int npt = max(1, min(2 + nbc / 3, (int)gridDim.x - 1));
    const int nbw = 0;      // (dedicated workgroups for the border-row chain TR(border, j) were measured too: the kernel ends 6 us after the last
                            //  diagonal block either way -- 575 us alone, 680 us beside the apply pass's transforms -- so they stay in the queue)
    int pt_next = ((int)blockIdx.x < npt) ? (int)blockIdx.x : nbc;
    int bd_next = ((int)blockIdx.x >= npt && (int)blockIdx.x < npt + nbw) ? (int)blockIdx.x - npt : nbc;
    for (;;) {
        __syncthreads();                    // s_task (and the LDS tiles) of the previous task are no longer read
        int kind = -1, bi = 0, bj = 0;      // kind 0: PT(bj); kind 1: TR(bi, bj)
        if (pt_next < nbc) { kind = 0; bj = pt_next; pt_next += npt; }      // (workgroup uniform)
        else if (bd_next < nbc) { kind = 1; bi = nbc; bj = bd_next; bd_next += nbw; }
        else {
            if (tid == 0) s_task = (int)atomicAdd(queue, 1u);
            __syncthreads();
            int t = s_task;
            // ---- decode: group j (1 <= j < nbc) = {TR(j+1..nbc-1, j-1)} (+ TR(border, j-1) when there are no border workgroups), last = {TR(border, nbc-1)}
            const int wb = nbw > 0 ? 0 : 1;
            int j = 1;
            for (; j < nbc; ++j) {
                const int cnt = nbc - j - 1 + wb;
                if (t < cnt) break;
                t -= cnt;
            }
            if (j < nbc) { kind = 1; bj = j - 1; bi = j + 1 + t; }           // t = 0 .. : rows j+1 .. nbc-1, then nbc (= border) if it is in the queue
            else if (t == 0 && wb) { kind = 1; bj = nbc - 1; bi = nbc; }
        }
        if (kind < 0) {                     // queue exhausted
            if (trace && tid == 0) atomicMax(&trace[12], wall_clock64());                           // (row 0, slot 12: the last workgroup's exit)
            return;
        }

        if (kind == 1) {
            // ================= TR(bi, bj): C = A(bi,bj) - sum_k L(bi,k) L(bj,k)^T;  X L(bj,bj)^T = C =================
            const int c0 = bj * CB, nbj = min(CB, n - c0);
            const int r0 = (bi == nbc) ? n : bi * CB;
            const int ni = (bi == nbc) ? 1 : min(CB, n - r0);
            d4s c[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 16 * wv + lk + 4 * q, jc = 16 * jt + ln;
                    const double v = A[(size_t)(r0 + min(i, ni - 1)) * ld + c0 + min(jc, nbj - 1)];
                    c[jt][q] = (i < ni && jc < nbj) ? v : 0.0;
                }
            for (int k = 0; k < bj; ++k) {
                if (tid == 0) { df_poll(flags, df_flag_id(g, bi, k), epoch, status); df_poll(flags, df_flag_id(g, bj, k), epoch, status); df_acquire(); }
                __syncthreads();            // (also: the previous iteration's MFMAs have read Li / Lj)
                df_load_tile(A, ld, r0, ni, k * CB, Li, tid);
                df_load_tile(A, ld, c0, nbj, k * CB, Lj, tid);
                __syncthreads();
#pragma unroll 4
                for (int ks = 0; ks < CB / 4; ++ks) {
                    const double av = -Li[16 * wv + ln][4 * ks + lk];
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) c[jt] = mfma16(av, Lj[16 * jt + ln][4 * ks + lk], c[jt]);
                }
            }
            __syncthreads();
            double (*T)[CB + 1] = Li;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) T[16 * wv + lk + 4 * q][16 * jt + ln] = c[jt][q];
            if (tid == 0) { df_poll(flags, df_flag_id(g, bj, bj), epoch, status); df_acquire(); }
            __syncthreads();
            {   // the factor of the diagonal block (identity beyond nbj) and the inverses of its 16 x 16 diagonal sub-blocks
                double dv[16], wvv[4];
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int e = tid + 256 * it, i = e >> 6, q = e & 63;
                    dv[it] = A[(size_t)(c0 + min(i, nbj - 1)) * ld + c0 + min(q, nbj - 1)];
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) wvv[it] = w16[(size_t)bj * 1024 + tid + 256 * it];
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int e = tid + 256 * it, i = e >> 6, q = e & 63;
                    L.Dl[i][q] = (i < nbj && q <= i) ? dv[it] : ((i == q) ? 1.0 : 0.0);
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) { const int e = tid + 256 * it; W16t[e >> 8][(e >> 4) & 15][e & 15] = wvv[it]; }
            }
            __syncthreads();
            df_trsm_mfma(T, L.Dl, W16t, wv, ln, lk);
            __syncthreads();
            df_store_tile(rsrc, ld, r0, ni, c0, T, tid);
            df_publish(flags, df_flag_id(g, bi, bj), epoch, tid);
            if (trace && tid == 0 && bi == nbc) trace[(size_t)bj * 16 + 10] = wall_clock64();       // (slot 10: border tile of column bj published)
            continue;
        }

        // ================= PT(bj): X = (bj, bj-1), D = (bj, bj) =================
        const int j = bj;
        const int d0 = j * CB, nbd = min(CB, n - d0);           // rows of both tiles, columns of D
        const int x0 = (j - 1) * CB;                             // columns of X (a full block when j >= 1)
        d4s cx[4], cd[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * wv + lk + 4 * q, jc = 16 * jt + ln;
                const int ir = d0 + min(i, nbd - 1);
                const double vd = A[(size_t)ir * ld + d0 + min(jc, nbd - 1)];
                const double vx = A[(size_t)ir * ld + (j > 0 ? x0 + jc : d0 + min(jc, nbd - 1))];      // (j = 0 has no X tile: any valid address)
                cd[jt][q] = (i < nbd && jc <= i) ? vd : 0.0;     // lower triangle of the diagonal tile
                cx[jt][q] = (j > 0 && i < nbd) ? vx : 0.0;
            }
        for (int k = 0; k + 1 < j; ++k) {
            if (tid == 0) { df_poll(flags, df_flag_id(g, j, k), epoch, status); df_poll(flags, df_flag_id(g, j - 1, k), epoch, status); df_acquire(); }
            __syncthreads();
            df_load_tile(A, ld, d0, nbd, k * CB, Li, tid);       // L(j, k)
            df_load_tile(A, ld, x0, CB, k * CB, Lj, tid);        // L(j-1, k)
            __syncthreads();
#pragma unroll 2
            for (int ks = 0; ks < CB / 4; ++ks) {
                const double av = -Li[16 * wv + ln][4 * ks + lk];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    cx[jt] = mfma16(av, Lj[16 * jt + ln][4 * ks + lk], cx[jt]);
                    cd[jt] = mfma16(av, Li[16 * jt + ln][4 * ks + lk], cd[jt]);
                }
            }
        }
        __syncthreads();
        DF_TRACE(0);
        double (*T)[CB + 1] = Li;
        if (j > 0) {
            // ---- critical path: X L(j-1,j-1)^T = X as soon as the factor of D(j-1) is out ----
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) T[16 * wv + lk + 4 * q][16 * jt + ln] = cx[jt][q];
            if (tid == 0) { df_poll(flags, df_flag_id(g, j - 1, j - 1), epoch, status); DF_TRACE(1); df_acquire(); }
            __syncthreads();
            DF_TRACE(2);
            {
                double dv[16], wvv[4];
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int e = tid + 256 * it, i = e >> 6, q = e & 63;
                    dv[it] = A[(size_t)(x0 + i) * ld + x0 + q];
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) wvv[it] = w16[(size_t)(j - 1) * 1024 + tid + 256 * it];
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int e = tid + 256 * it, i = e >> 6, q = e & 63;
                    L.Dl[i][q] = (q <= i) ? dv[it] : 0.0;
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) { const int e = tid + 256 * it; W16t[e >> 8][(e >> 4) & 15][e & 15] = wvv[it]; }
            }
            __syncthreads();
            DF_TRACE(3);
            df_trsm_mfma(T, L.Dl, W16t, wv, ln, lk);                         // X: published, and the operand tile of D -= X X^T (rows >= nbd stay zero)
            DF_TRACE(4);
            __syncthreads();
            df_store_tile(rsrc, ld, d0, nbd, x0, T, tid);
            DF_TRACE(5);
            // (the X tile's stores drain while the matrix instructions of D -= X X^T issue; its flag goes out after them: the consumers
            //  of X -- accumulations of later tasks -- have more than a block step of slack, the chain has none)
#pragma unroll 4
            for (int ks = 0; ks < CB / 4; ++ks) {
                const double av = -T[16 * wv + ln][4 * ks + lk];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) cd[jt] = mfma16(av, T[16 * jt + ln][4 * ks + lk], cd[jt]);
            }
            df_publish(flags, df_flag_id(g, j, j - 1), epoch, tid);      // (its barrier also ends the reads of T)
        }
        // ---- factor D ----
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) T[16 * wv + lk + 4 * q][16 * jt + ln] = cd[jt][q];
        __syncthreads();
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cc = cg + 4 * q;
            a[q] = (ti < nbd && cc <= ti) ? T[ti][cc] : ((ti == cc) ? 1.0 : 0.0);           // identity padding beyond nbd
        }
        __syncthreads();                    // T has been read; the PanelLds region may be overwritten
        DF_TRACE(6);
        chol_factor_diag<true>(a, L, tid, nbd, true, status, Vd);
        DF_TRACE(9);
        {   // inverses of the 16 x 16 diagonal sub-blocks, one per wave, published with the factor: lane = 4 c + h stores
            // W16t[wv][c][4 h .. 4 h + 3] = 32 contiguous bytes of w16[j][wv][c][*]
            chol_inv16_mfma(L.Dl, Vd, reinterpret_cast<double (*)[16][W16_LD]>(&T[0][0]) + 3 * wv, W16t, wv, ln, lk);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double v0 = W16t[wv][lane >> 2][4 * (lane & 3) + 2 * h], v1 = W16t[wv][lane >> 2][4 * (lane & 3) + 2 * h + 1];
                df_u4 pk;
                pk.x = (unsigned int)__double2loint(v0); pk.y = (unsigned int)__double2hiint(v0);
                pk.z = (unsigned int)__double2loint(v1); pk.w = (unsigned int)__double2hiint(v1);
                __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc_w, (int)(((size_t)j * 1024 + (size_t)wv * 256 + (size_t)lane * 4 + 2 * h) * sizeof(double)), 0, /*aux: sc1*/ 16);
            }
        }
        DF_TRACE(7);
        df_store_tile(rsrc, ld, d0, nbd, d0, L.Dl, tid);                     // (Dl is zero above the diagonal)
        if (tid < nbd) __hip_atomic_store(&rd[d0 + tid], L.rdiag[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        df_publish(flags, df_flag_id(g, j, j), epoch, tid);
        DF_TRACE(8);
        // off the critical path (the factor is out): the full inverse of the diagonal block for the back substitution
        chol_inv64_mfma(L.Dl, W16t, T, Sm64, tid, wv, ln, lk);
        {
            double* Wg = winv + (size_t)j * CB * CB;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int e = tid + 256 * it;
                Wg[e] = T[e >> 6][e & 63];
            }
        }
    }
#undef DF_TRACE
}

// ---- the four panel steps of one 256-column outer block as ONE launch ------------------------------------------------------
// (outer-blocked factorisation of large systems, below).  One workgroup per 64-row tile of the block column, rows kb .. n (the border
// row rides in the last tile).  Roles are handed out by an atomic counter in start order, so a workgroup only ever waits for workgroups
// that are already running: role r owns row tile r.  It keeps its four 64 x 64 tiles (columns kb + 64 st, st = 0 .. 3) in the matrix
// accumulators and walks the steps:
//   r == st: factor the diagonal tile, publish the factor and the inverses of its 16 x 16 diagonal sub-blocks (flag st), done;
//   r  > st: wait for that flag, X = T_st L^-T on the matrix cores (df_trsm_mfma), store X (final values of the factor; roles 1 .. 3
//            also publish it: flag 4 + 4 st + r), then T_sp -= X X_sp^T for the later tiles sp = st + 1 .. min(3, r) of its row, with
//            X_sp = role sp's X tile of this step (its own when sp == r).
// Critical path per step: flag -> load factor -> solve -> update the next diagonal tile -> factor (13 us) -> publish: ~18 - 20 us
// against the 35 - 40 us of a chol_step launch (whose every column-0 workgroup factors the diagonal block redundantly, behind a
// launch boundary).  Hand-offs follow chol_dataflow's recipe (write-through payload, drained, one relaxed flag store; one poller,
// one agent-scope acquire).  The stamp carries the solve counter and the outer block, so the 16 flags are reused without a reset.
__global__ void __launch_bounds__(256) chol_panel4(double* __restrict__ A, int ld, int n, int kb, unsigned int* __restrict__ pflags,
                                                   unsigned int* __restrict__ role_ctr, const unsigned int* __restrict__ epoch_ctr,
                                                   unsigned int outer_id, int* __restrict__ status, double* __restrict__ rd,
                                                   double* __restrict__ w16, int nbc)
{
    __shared__ double smem[2 * CB * (CB + 1) + CB + 16 * CB + 16 + 4 * 16 * W16_LD + 4 * CB];
    __shared__ int s_role;
    double (*T)[CB + 1] = reinterpret_cast<double (*)[CB + 1]>(smem);                        // this step's tile / X
    double (*Lj)[CB + 1] = reinterpret_cast<double (*)[CB + 1]>(smem + CB * (CB + 1));       // another role's X tile / start of the PanelLds
    PanelLds& L = *reinterpret_cast<PanelLds*>(smem + CB * (CB + 1));
    double (*W16t)[16][W16_LD] = reinterpret_cast<double (*)[16][W16_LD]>(smem + 2 * CB * (CB + 1) + CB + 16 * CB + 16);
    double (*Vd)[4] = reinterpret_cast<double (*)[4]>(smem + 2 * CB * (CB + 1) + CB + 16 * CB + 16 + 4 * 16 * W16_LD);
    const unsigned int epoch = (*epoch_ctr << 8) | (outer_id & 0xFFu);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, lk = lane >> 4;
    const int ti = 16 * wv + ln, cg = lk;                    // element ownership of chol_factor_diag<true>
    if (tid == 0) s_role = (int)atomicAdd(role_ctr + outer_id, 1u);
    __syncthreads();
    const int r = s_role;
    const int R0 = kb + CB * r, nr = min(CB, n + 1 - R0);
    if (nr <= 0) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((size_t)(n + 1) * ld * sizeof(double)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)w16, 0, (int)((size_t)nbc * 1024 * sizeof(double)), 0x00020000);
    // my four tiles, in the accumulator layout C[16 wv + lk + 4 q][16 jt + ln] (tiles right of the diagonal of rows < 256 are never used)
    d4s cT[4][4];
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * wv + lk + 4 * q, jc = 16 * jt + ln;
                const double v = A[(size_t)(R0 + min(i, nr - 1)) * ld + kb + CB * st + jc];
                cT[st][jt][q] = (i < nr && (st != r || jc <= i)) ? v : 0.0;
            }
    auto step = [&](auto ST) -> bool {                       // returns true when this role is finished
        constexpr int st = decltype(ST)::value;
        if (r < st) return true;
        __syncthreads();                                     // (T / Lj of the previous step are no longer read)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) T[16 * wv + lk + 4 * q][16 * jt + ln] = cT[st][jt][q];
        if (r == st) {
            // ---- factor the diagonal tile (a full 64 x 64 block: the host only sends outer blocks with >= 320 columns left) ----
            __syncthreads();
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int cc = cg + 4 * q;
                a[q] = (cc <= ti) ? T[ti][cc] : 0.0;
            }
            __syncthreads();                                 // T has been read; the PanelLds region may be overwritten
            chol_factor_diag<true>(a, L, tid, CB, true, status, Vd);
            chol_inv16_mfma(L.Dl, Vd, reinterpret_cast<double (*)[16][W16_LD]>(&T[0][0]) + 3 * wv, W16t, wv, ln, lk);
            const int jb = kb / CB + st;                     // block index of this diagonal tile (w16 slot)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double v0 = W16t[wv][lane >> 2][4 * (lane & 3) + 2 * h], v1 = W16t[wv][lane >> 2][4 * (lane & 3) + 2 * h + 1];
                df_u4 pk;
                pk.x = (unsigned int)__double2loint(v0); pk.y = (unsigned int)__double2hiint(v0);
                pk.z = (unsigned int)__double2loint(v1); pk.w = (unsigned int)__double2hiint(v1);
                __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc_w, (int)(((size_t)jb * 1024 + (size_t)wv * 256 + (size_t)lane * 4 + 2 * h) * sizeof(double)), 0, /*aux: sc1*/ 16);
            }
            df_store_tile(rsrc, ld, R0, CB, kb + CB * st, L.Dl, tid);          // (Dl is zero above the diagonal)
            if (tid < CB) __hip_atomic_store(&rd[R0 + tid], L.rdiag[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            df_publish(pflags, st, epoch, tid);
            return true;
        }
        // ---- r > st: X = T L(st,st)^-T ----
        if (tid == 0) { df_poll(pflags, st, epoch, status); df_acquire(); }
        __syncthreads();
        {
            const int c0 = kb + CB * st, jb = kb / CB + st;
            double dv[16], wvv[4];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int e = tid + 256 * it, i = e >> 6, q = e & 63;
                dv[it] = A[(size_t)(c0 + i) * ld + c0 + q];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) wvv[it] = w16[(size_t)jb * 1024 + tid + 256 * it];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int e = tid + 256 * it, i = e >> 6, q = e & 63;
                L.Dl[i][q] = (q <= i) ? dv[it] : 0.0;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) { const int e = tid + 256 * it; W16t[e >> 8][(e >> 4) & 15][e & 15] = wvv[it]; }
        }
        __syncthreads();
        df_trsm_mfma(T, L.Dl, W16t, wv, ln, lk);
        __syncthreads();
        df_store_tile(rsrc, ld, R0, nr, kb + CB * st, T, tid);
        if (r <= 3) df_publish(pflags, 4 + 4 * st + r, epoch, tid);
        // ---- updates of my later tiles ----
        auto upd = [&](auto SP) {
            constexpr int sp = decltype(SP)::value;
            if constexpr (sp > st && sp <= 3) {
                if (sp > r) return;
                const double (*B)[CB + 1] = T;
                if (sp != r) {
                    __syncthreads();                         // (the trsm's L.Dl / the previous sp's tile in Lj have been read)
                    if (tid == 0) { df_poll(pflags, 4 + 4 * st + sp, epoch, status); df_acquire(); }
                    __syncthreads();
                    df_load_tile(A, ld, kb + CB * sp, CB, kb + CB * st, Lj, tid);
                    __syncthreads();
                    B = Lj;
                }
#pragma unroll 4
                for (int ks = 0; ks < CB / 4; ++ks) {
                    const double av = -T[16 * wv + ln][4 * ks + lk];
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) cT[sp][jt] = mfma16(av, B[16 * jt + ln][4 * ks + lk], cT[sp][jt]);
                }
            }
        };
        upd(std::integral_constant<int, 1>{}); upd(std::integral_constant<int, 2>{}); upd(std::integral_constant<int, 3>{});
        return false;
    };
    if (step(std::integral_constant<int, 0>{})) return;
    if (step(std::integral_constant<int, 1>{})) return;
    if (step(std::integral_constant<int, 2>{})) return;
    step(std::integral_constant<int, 3>{});
}

// ---- outer blocking for large systems ----------------------------------------------------------------------------
// With rank-64 updates every step reads and writes the whole trailing matrix: n^3 / 24 bytes in total (31 GB at n = 7231,
// 12 ms).  For large n the factorisation therefore works on outer blocks of 256 columns: the four inner steps only update
// the columns of their own outer block, and the rest of the trailing matrix receives all four panels at once from this
// kernel: C -= L_K L_K^T with K = 256, 64 x 64 tiles on the matrix cores (one 32 x 32 quadrant per wave, four 16 x 16
// accumulators per lane), the panels staged through LDS sixteen columns at a time with the next chunk's loads in flight.
#define SYRK_KC 16
#define SYRK_S (SYRK_KC + 4)
// tj0: first tile column of this launch.  TS = tile side.  Rounds 2 - 3 ran 128 x 128 tiles (half the LDS reads per matrix
// instruction): such a tile alone on a CU takes 62 us however few of them a launch has, two sharing a CU 90 us, and only two fit.
// 64 x 64 tiles (20 KB of LDS, eight workgroups per CU) are faster for EVERY outer block of configs 3 and 5: solve 7.42 -> 6.29 ms at
// n = 7207, 6.18 -> 5.42 ms at n = 6251 (round 4; thresholds in between were in between).
template <int TS>
__global__ void __launch_bounds__(256, 2) chol_syrk(double* __restrict__ A, int ld, int n, int K0, int KB, int r0, int tj0)
{
    constexpr int NQ = TS / 32;             // 16 x 16 accumulators per wave and dimension (the wave's quadrant is TS / 2 square)
    constexpr int NU = TS / 32;             // staging rows per thread
    const int ti = blockIdx.y, tj = blockIdx.x + tj0;
    if (tj > ti) return;
    __shared__ double Li[TS][SYRK_S];
    __shared__ double Lj[TS][SYRK_S];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, lk = lane >> 4;
    const int wr = wv >> 1, wc = wv & 1;
    const int i0 = r0 + ti * TS, j0 = r0 + tj * TS;
    // staging: thread loads rows (tid >> 3) + 32 u, u < NU, columns 2 (tid & 7) .. + 1 of the chunk, for both panels
    const int sr = tid >> 3, sc = 2 * (tid & 7);
    double2 pi[NU], pj[NU];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int ri = i0 + sr + 32 * u, rj = j0 + sr + 32 * u;
            const double2 qi = *reinterpret_cast<const double2*>(A + (size_t)min(ri, n) * ld + K0 + kc + sc);      // (clamped, then masked)
            const double2 qj = *reinterpret_cast<const double2*>(A + (size_t)min(rj, n) * ld + K0 + kc + sc);
            pi[u] = (ri <= n) ? qi : make_double2(0.0, 0.0);
            pj[u] = (rj < n) ? qj : make_double2(0.0, 0.0);
        }
    };
    d4s c[NQ][NQ];
#pragma unroll
    for (int it = 0; it < NQ; ++it)
#pragma unroll
        for (int jt = 0; jt < NQ; ++jt) c[it][jt] = (d4s){0.0, 0.0, 0.0, 0.0};
    fetch(0);
    for (int kc = 0; kc < KB; kc += SYRK_KC) {
        __syncthreads();                    // the previous chunk has been consumed
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            Li[sr + 32 * u][sc] = pi[u].x; Li[sr + 32 * u][sc + 1] = pi[u].y;
            Lj[sr + 32 * u][sc] = pj[u].x; Lj[sr + 32 * u][sc + 1] = pj[u].y;
        }
        __syncthreads();
        if (kc + SYRK_KC < KB) fetch(kc + SYRK_KC);
#pragma unroll
        for (int ks = 0; ks < SYRK_KC / 4; ++ks) {
            double a[NQ], b[NQ];
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                a[t] = Li[(TS / 2) * wr + 16 * t + ln][4 * ks + lk];
                b[t] = Lj[(TS / 2) * wc + 16 * t + ln][4 * ks + lk];
            }
#pragma unroll
            for (int it = 0; it < NQ; ++it)
#pragma unroll
                for (int jt = 0; jt < NQ; ++jt) c[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[it], b[jt], c[it][jt], 0, 0, 0);
        }
    }
    // read-modify-write of the tile, 4 NQ elements at a time: all loads first (written as `A[..] -= c` the compiler orders
    // every load behind the previous store -- 64 dependent round trips per thread)
#pragma unroll
    for (int it = 0; it < NQ; ++it) {
        double oldv[NQ][4];
#pragma unroll
        for (int jt = 0; jt < NQ; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + (TS / 2) * wr + 16 * it + (lk + 4 * q), j = j0 + (TS / 2) * wc + 16 * jt + ln;
                oldv[jt][q] = (i <= n && j < n && j <= i) ? A[(size_t)i * ld + j] : 0.0;
            }
#pragma unroll
        for (int jt = 0; jt < NQ; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + (TS / 2) * wr + 16 * it + (lk + 4 * q), j = j0 + (TS / 2) * wc + 16 * jt + ln;
                if (i <= n && j < n && j <= i) A[(size_t)i * ld + j] = oldv[jt][q] - c[it][jt][q];
            }
    }
}

// ---- back substitution in two launches -----------------------------------------------------------------------
// L^T x = y, y = border row of the factor.  A launch per 64-column block (the version above) costs ~20 us per block, almost
// all of it launch and hand-off latency: 28 blocks -> 0.55 ms at n = 1735.  Instead:
//  (1) chol_inv_diag: W_b = L_bb^-1 for every diagonal block, all blocks in parallel;
//  (2) chol_back_all: one resident workgroup per block b, x_b = W_b^T (y_b - sum_{c > b} L_cb^T x_c).  Workgroup b folds
//      in the blocks x_c as they are published (flag per block, epoch-stamped so that nothing needs zeroing), last to first;
//      the only work on the critical path per block is two 64 x 64 matrix-vector products out of LDS and one flag hand-off.
__global__ void __launch_bounds__(256) chol_inv_diag(const double* __restrict__ A, int ld, int n, const double* __restrict__ rd,
                                                     double* __restrict__ Winv)
{
    __shared__ double Ls[CB][CB + 1];
    __shared__ double Ws[CB][CB + 1];
    const int tid = threadIdx.x, kb = blockIdx.x * CB, nb = min(CB, n - kb);
    {   // all 16 loads (clamped), a scheduling fence, then the masked stores: in one loop the compiler sinks each load back into
        // a branch of its own with a full wait behind it
        double lv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it, r = e >> 6, q = e & 63;
            lv[it] = A[(size_t)(kb + min(r, nb - 1)) * ld + kb + min(q, nb - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it, r = e >> 6, q = e & 63;
            Ls[r][q] = (r < nb && q <= r) ? lv[it] : 0.0;
            Ws[r][q] = 0.0;
        }
    }
    __syncthreads();
    // column j of W solves L w = e_j; four lanes per column split the dot product (same wave: tid = 4 j + g)
    const int j = tid >> 2, g = tid & 3;
    for (int i = 0; i < nb; ++i) {
        double sacc = 0.0;
        if (j <= i) for (int k = j + g; k < i; k += 4) sacc = fma(Ls[i][k], Ws[k][j], sacc);
        sacc += __shfl_xor(sacc, 1);
        sacc += __shfl_xor(sacc, 2);
        if (g == 0 && j <= i && j < nb) Ws[i][j] = ((i == j ? 1.0 : 0.0) - sacc) * rd[kb + i];
        __syncthreads();
    }
    double* W = Winv + (size_t)blockIdx.x * CB * CB;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, r = e >> 6, q = e & 63;
        W[e] = Ws[r][q];
    }
}

__global__ void __launch_bounds__(256) chol_back_all(const double* __restrict__ A, int ld, int n, const double* __restrict__ Winv,
                                                     double* xv, unsigned int* flags, const unsigned int* __restrict__ epoch_ctr,
                                                     int* __restrict__ status)
{
    const unsigned int epoch = (*epoch_ctr << 8) | 0xFFu;         // (step ids of chol_step stay below 255)
    __shared__ double Wb[CB][CB + 1];
    __shared__ double Ln[CB][CB + 1];
    __shared__ double xs[CB];
    __shared__ double red[4][CB];
    const int nblk = gridDim.x;
    const int b = nblk - 1 - (int)blockIdx.x;          // the last block has no dependency: give it the first workgroup
    const int tid = threadIdx.x, kb = b * CB, nb = min(CB, n - kb);
    const int jj = tid & 63, gg = tid >> 6;
    const double* W = Winv + (size_t)b * CB * CB;
    {   // (loads first, a scheduling fence, then the masked stores: see chol_inv_diag)
        double wv[16], lv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it, r = e >> 6, q = e & 63;
            wv[it] = W[e];
            lv[it] = A[(size_t)min(kb + CB + r, n) * ld + kb + min(q, nb - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it, r = e >> 6, q = e & 63;
            Wb[r][q] = wv[it];
            Ln[r][q] = (b + 1 < nblk && kb + CB + r < n && q < nb) ? lv[it] : 0.0;
        }
    }
    double acc = 0.0;                                   // threads tid < 64: sum_c (L_cb^T x_c)[tid]
    const double yb = (tid < nb) ? A[(size_t)n * ld + kb + tid] : 0.0;      // (fetched before the chain, not on it)
    for (int c = nblk - 1; c > b; --c) {
        if (tid == 0) { df_poll(flags, c, epoch, status); df_acquire(); }      // relaxed polls, one acquire (bounded: reports, never hangs)
        __syncthreads();
        const int kc = c * CB, nc = min(CB, n - kc);
        if (tid < CB) xs[tid] = tid < nc ? __hip_atomic_load(&xv[kc + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        __syncthreads();
        double ps = 0.0;
        if (c == b + 1) {
#pragma unroll 4
            for (int i = gg; i < CB; i += 4) ps = fma(Ln[i][jj], xs[i], ps);
        } else if (jj < nb) {
#pragma unroll 4
            for (int i = gg; i < nc; i += 4) ps = fma(A[(size_t)(kc + i) * ld + kb + jj], xs[i], ps);
        }
        red[gg][jj] = ps;
        __syncthreads();
        if (tid < CB) acc += red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
    __syncthreads();
    if (tid < CB) xs[tid] = tid < nb ? yb - acc : 0.0;                                  // y_b - strip products
    __syncthreads();
    double ps = 0.0;
#pragma unroll 4
    for (int i = gg; i < CB; i += 4) ps = fma(Wb[i][jj], xs[i], ps);                   // x_b = W_b^T rhs
    red[gg][jj] = ps;
    __syncthreads();
    // publish as chol_dataflow does: write-through stores by the first wave, drained, then one relaxed flag store by its first lane
    // (a fence + release store here cost ~1 us of the 3.5 us a block spends on the chain)
    if (tid < CB) {
        if (tid < nb) __hip_atomic_store(&xv[kb + tid], red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) __hip_atomic_store(&flags[b], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void __launch_bounds__(256) scatter_solution(const double* __restrict__ xv, int n, const int* __restrict__ idx,
                                                        double* __restrict__ solution, int NEQ, int tie_first, int tie_cnt, int tie_stride)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) solution[idx ? idx[t] : t] = xv[t];
    if (t >= 1 && t < tie_cnt) solution[tie_first + t * tie_stride] = xv[tie_first];   // position of tie_first in x equals its value
}

// (the pivoted LU lives in lu.hpp)

#endif
