// fill.hpp -- FillLS_* / stripe removal / tied scaling: one thread per matrix element.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_FILL_HPP
#define SFFT_AMD_FILL_HPP

// ------------------------------------------------------------------------------------------------
// FillLS_{OMG,GAM,PSI,PHI,THE,DEL} + Remove_LSFStripes, one thread per matrix element
// (SFFTConfigure.py:957-1293).  PSI is filled from GAM through Psi[p'q',ij](-rho) == Gam[ij,p'q'](rho), and
// Omega pairs with i'j' > ij from Omega[ij,i'j'](-rho); both identities are exact.
// out is [(n+1)][ld]: rows/cols < n hold LHMAT (after the optional index map), row n and column n hold RHb.
// ------------------------------------------------------------------------------------------------
struct FillArgs {
    int Fij, Fpq, Fab, Fijab, L1, w0, w1;
    int h_omg;            // 2w
    int h_gam;            // w
    int omg_off;          // patches offset of Omega pair 0; pair (i'<=i) index = i'*Fij - i'(i'-1)/2 + (i - i')
    int gam_off;          // Gam pair (ij, pq) at gam_off + (ij*Fpq+pq)*PHg*PHg
    int the_off;          // Theta pair ij at the_off + ij*PHg*PHg
    // tied scaling (B-spline constant photometric ratio, BSplineSFFT.py:2201-2272): the unknowns tie_first + k*tie_stride,
    // k < tie_cnt, are one unknown; its row / column is the SUM of theirs.  tie_cnt = 0: no tie.
    int tie_first, tie_cnt, tie_stride;
    // separately varying scaling (BSplineSFFT.py SCALING_MODE 'SEPARATE-VARYING', :1348-2005): the centre unknown
    // (ij, 00) belongs to SCALING plane ij (a zero place-holder for ij >= nsca) instead of kernel plane ij.  Patches of
    // half width h_gam: sk (s, ij) = ScaI_s x I_ij, ss (s <= t) = ScaI_s x ScaI_t, sg (s, pq), st (s) = ScaI_s x J.
    int sv, nsca;
    int sk_off, ss_off, sg_off, st_off;
    // kernel regularisation (BSplineSFFT.py:2007-2168, 3570-3700): LHMAT += reg_coef * S[k][k8] * ireg[c][c8], with
    // S = sst, or for separately varying scaling csst / dsst when one / both of c, c8 are the kernel centre.
    double reg_coef;                 // LAMBDA_REGULARIZE * SCALE^2; 0 = off
    const double* ireg;              // [Fab][Fab]
    const double *sst, *csst, *dsst; // [Fij][Fij]
};

__device__ __forceinline__ double omg_at(const double* P, const FillArgs& f, int i8, int ij, int r0, int r1)
{
    const int PH = 2 * f.h_omg + 1;
    int lo = i8, hi = ij;
    if (i8 > ij) { lo = ij; hi = i8; r0 = -r0; r1 = -r1; }
    const int pidx = lo * f.Fij - (lo * (lo - 1)) / 2 + (hi - lo);
    return P[f.omg_off + (size_t)pidx * PH * PH + (size_t)(r0 + f.h_omg) * PH + (r1 + f.h_omg)];
}

// lag patches of the scaling planes (h_gam wide)
__device__ __forceinline__ double sk_at(const double* P, const FillArgs& f, int s, int ij, int r0, int r1)
{
    if (s >= f.nsca) return 0.0;
    const int PHg = 2 * f.h_gam + 1;
    return P[f.sk_off + (size_t)(s * f.Fij + ij) * PHg * PHg + (size_t)(r0 + f.h_gam) * PHg + (r1 + f.h_gam)];
}
__device__ __forceinline__ double ss_at0(const double* P, const FillArgs& f, int s, int t)
{
    if (s >= f.nsca || t >= f.nsca) return 0.0;
    const int PHg = 2 * f.h_gam + 1;
    const int lo = min(s, t), hi = max(s, t);
    const int pidx = lo * f.nsca - (lo * (lo - 1)) / 2 + (hi - lo);
    return P[f.ss_off + (size_t)pidx * PHg * PHg + (size_t)f.h_gam * PHg + f.h_gam];     // lag 0 is symmetric in (s, t)
}

__device__ double reg_element(const FillArgs& f, int i8, int ab8, bool c8, int ij, int ab, bool c)
{
    const double* S = f.sst;
    if (f.sv) {
        if (c8 && c) S = f.dsst;
        else if (c) return f.reg_coef * f.csst[i8 * f.Fij + ij] * f.ireg[(size_t)ab8 * f.Fab + ab];
        else if (c8) return f.reg_coef * f.csst[ij * f.Fij + i8] * f.ireg[(size_t)ab8 * f.Fab + ab];
    }
    return f.reg_coef * S[i8 * f.Fij + ij] * f.ireg[(size_t)ab8 * f.Fab + ab];
}

__device__ double sys_element(const double* P, const double* phi, const double* delta, const FillArgs& f, int R, int C, int NEQ)
{
    const int PHg = 2 * f.h_gam + 1;
    if (C == NEQ) {   // right hand side
        if (R < f.Fijab) {
            const int i8 = R / f.Fab, ab8 = R - i8 * f.Fab;
            const int a8 = ab8 / f.L1 - f.w0, b8 = ab8 % f.L1 - f.w1;
            if (f.sv && a8 == 0 && b8 == 0)
                return i8 < f.nsca ? P[f.st_off + (size_t)i8 * PHg * PHg + (size_t)f.h_gam * PHg + f.h_gam] : 0.0;
            const double* T = P + f.the_off + (size_t)i8 * PHg * PHg;
            const double t0 = T[(size_t)f.h_gam * PHg + f.h_gam];
            if (a8 == 0 && b8 == 0) return t0;
            return T[(size_t)(a8 + f.h_gam) * PHg + (b8 + f.h_gam)] - t0;
        }
        return delta[R - f.Fijab];
    }
    if (R < f.Fijab && C < f.Fijab) {
        const int i8 = R / f.Fab, ab8 = R - i8 * f.Fab;
        const int ij = C / f.Fab, ab = C - ij * f.Fab;
        const int a8 = ab8 / f.L1 - f.w0, b8 = ab8 % f.L1 - f.w1;
        const int a = ab / f.L1 - f.w0, b = ab % f.L1 - f.w1;
        const bool c8 = (a8 == 0 && b8 == 0), c = (a == 0 && b == 0);
        const double reg = (f.reg_coef != 0.0) ? reg_element(f, i8, ab8, c8, ij, ab, c) : 0.0;
        if (f.sv && (c8 || c)) {
            // OMG01 = ScaI_i8 x I_ij read at (-a, -b); OMG10 = I_i8 x ScaI_ij at (a8, b8) == ScaI_ij x I_i8 at (-a8, -b8)
            if (c8 && c) return ss_at0(P, f, i8, ij) + reg;
            if (c8) return sk_at(P, f, i8, ij, -a, -b) - sk_at(P, f, i8, ij, 0, 0) + reg;
            return sk_at(P, f, ij, i8, -a8, -b8) - sk_at(P, f, ij, i8, 0, 0) + reg;
        }
        // the four-case centre rule without branches around the loads (all four lags are always inside the patch): the loads
        // issue together and the case picks the coefficients
        const double o00 = omg_at(P, f, i8, ij, 0, 0);
        const double o1 = omg_at(P, f, i8, ij, a8, b8);
        const double o2 = omg_at(P, f, i8, ij, -a, -b);
        const double o3 = omg_at(P, f, i8, ij, a8 - a, b8 - b);
        const bool none = !c8 && !c;
        const double k1 = none ? -1.0 : ((c && !c8) ? 1.0 : 0.0);
        const double k2 = none ? -1.0 : ((c8 && !c) ? 1.0 : 0.0);
        const double k3 = none ? 1.0 : 0.0;
        const double k0 = (none || (c8 && c)) ? 1.0 : -1.0;
        return fma(k1, o1, fma(k2, o2, fma(k3, o3, k0 * o00))) + reg;
    }
    if (R < f.Fijab) {          // GAM block
        const int pq = C - f.Fijab;
        const int i8 = R / f.Fab, ab8 = R - i8 * f.Fab;
        const int a8 = ab8 / f.L1 - f.w0, b8 = ab8 % f.L1 - f.w1;
        if (f.sv && a8 == 0 && b8 == 0)
            return i8 < f.nsca ? P[f.sg_off + (size_t)(i8 * f.Fpq + pq) * PHg * PHg + (size_t)f.h_gam * PHg + f.h_gam] : 0.0;
        const double* G = P + f.gam_off + (size_t)(i8 * f.Fpq + pq) * PHg * PHg;
        const double g0 = G[(size_t)f.h_gam * PHg + f.h_gam];
        if (a8 == 0 && b8 == 0) return g0;
        return G[(size_t)(a8 + f.h_gam) * PHg + (b8 + f.h_gam)] - g0;
    }
    if (C < f.Fijab) {          // PSI block = GAM transposed
        const int pq = R - f.Fijab;
        const int ij = C / f.Fab, ab = C - ij * f.Fab;
        const int a = ab / f.L1 - f.w0, b = ab % f.L1 - f.w1;
        if (f.sv && a == 0 && b == 0)
            return ij < f.nsca ? P[f.sg_off + (size_t)(ij * f.Fpq + pq) * PHg * PHg + (size_t)f.h_gam * PHg + f.h_gam] : 0.0;
        const double* G = P + f.gam_off + (size_t)(ij * f.Fpq + pq) * PHg * PHg;
        const double g0 = G[(size_t)f.h_gam * PHg + f.h_gam];
        if (a == 0 && b == 0) return g0;
        return G[(size_t)(a + f.h_gam) * PHg + (b + f.h_gam)] - g0;
    }
    return phi[(R - f.Fijab) * f.Fpq + (C - f.Fijab)];
}

// element of the (possibly tied) system: sum over the members of the row group and of the column group
__device__ double sys_group_element(const double* P, const double* phi, const double* delta, const FillArgs& f, int R, int C, int NEQ)
{
    const int nr = (f.tie_cnt && R == f.tie_first) ? f.tie_cnt : 1;
    const int nc = (f.tie_cnt && C == f.tie_first) ? f.tie_cnt : 1;
    double acc = 0.0;
    for (int a = 0; a < nr; ++a)
        for (int b = 0; b < nc; ++b) acc += sys_element(P, phi, delta, f, R + a * f.tie_stride, C + b * f.tie_stride, NEQ);
    return acc;
}

__global__ void __launch_bounds__(256) fill_system(const double* __restrict__ P, const double* __restrict__ phi,
                                                   const double* __restrict__ delta, FillArgs f, const int* __restrict__ idx,
                                                   int n, int NEQ, double* __restrict__ out, int ld,
                                                   double* __restrict__ rhs_vec, int lower_only)
{
    const int Cp = blockIdx.x * 16 + (threadIdx.x & 15);
    const int Rp = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (Rp > n || Cp > n) return;
    if (lower_only && Cp > Rp) return;          // the Cholesky path reads the lower triangle and the border row only
    if (Rp == n && Cp == n) { if (out) out[(size_t)n * ld + n] = 0.0; return; }
    if (Rp == n) {   // rhs row (and optional separate vector)
        const int C = idx ? idx[Cp] : Cp;
        const double v = sys_group_element(P, phi, delta, f, C, NEQ, NEQ);
        if (out) out[(size_t)n * ld + Cp] = v;
        if (rhs_vec) rhs_vec[Cp] = v;
        return;
    }
    if (!out) return;
    const int R = idx ? idx[Rp] : Rp;
    if (Cp == n) { out[(size_t)Rp * ld + n] = sys_group_element(P, phi, delta, f, R, NEQ, NEQ); return; }
    const int C = idx ? idx[Cp] : Cp;
    out[(size_t)Rp * ld + Cp] = sys_group_element(P, phi, delta, f, R, C, NEQ);
}

// plain LHMAT export for sfft_get_system (no border)
__global__ void __launch_bounds__(256) fill_plain(const double* __restrict__ P, const double* __restrict__ phi,
                                                  const double* __restrict__ delta, FillArgs f, int NEQ,
                                                  double* __restrict__ LH, double* __restrict__ rhs)
{
    const int C = blockIdx.x * 16 + (threadIdx.x & 15);
    const int R = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (R >= NEQ || C >= NEQ) return;
    if (LH) LH[(size_t)R * NEQ + C] = sys_element(P, phi, delta, f, R, C, NEQ);
    if (rhs && C == 0) rhs[R] = sys_element(P, phi, delta, f, R, NEQ, NEQ);
}

#endif
