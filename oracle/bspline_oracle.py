"""CPU oracle for the B-spline form of SFFT -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/scipy restatement of the reference's B-spline-capable Numpy backend

    misc/beta4spline/new_version_sfftcore/SFFTConfigure.py:1032-1768   SingleSFFTConfigure_Numpy.SSCN
    misc/beta4spline/new_version_sfftcore/SFFTSubtract.py:557-971     ElementalSFFTSubtract_Numpy.ESSN
    misc/beta4spline/new_version_sfftcore/SFFTSubtract.py:989-1040    GeneralSFFTSubtract.GSS

which is the only CPU implementation the reference has of the algorithm in sfft/BSplineSFFT.py (that file is
CuPy-only).  It covers BSplineSFFT's scaling modes ENTANGLED (ConstPhotRatio=False) and SEPARATE-CONSTANT
(ConstPhotRatio=True; same TweakLS rule: BSplineSFFT.py:3707-3768 vs SFFTConfigure.py:1615-1699 of the dev version)
without kernel regularisation.  SEPARATE-VARYING scaling and regularisation have no CPU implementation in the
reference; they are restated in oracle/bspline_sv_oracle.py (pinned since round 3 through the reference's NIRCam golden, see there).

Parity status: PINNED by tests/golden/bs_*.npz (tests/golden/make_golden_bspline.py imports the dev-version modules
in the build container); tests/test_oracle_golden.py checks this file against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
from scipy.interpolate import BSpline

from .sfft_oracle import _fft2, _ifft2


def create_bspline_basis(N, IntKnot, BSplineDegree):
    """SFFTSubtract.py:565-576 of the dev version == sfft/BSplineSFFT.py:2624-2645: basis functions on the scaled
    pixel-centre coordinates (1..N)/N with boundary knots at 0.5/N and (N+0.5)/N."""
    PixCoord = (1.0 + np.arange(N)) / N
    Knot = np.concatenate(([0.5] * (BSplineDegree + 1), IntKnot, [N + 0.5] * (BSplineDegree + 1))) / N
    Nc = len(IntKnot) + BSplineDegree + 1
    out = []
    for idx in range(Nc):
        Coeff = (np.arange(Nc) == idx).astype(float)
        out.append(BSpline(t=Knot, c=Coeff, k=BSplineDegree, extrapolate=False)(PixCoord))
    return np.array(out).astype(np.float64)


def make_basis(N0, N1, KerSpType='Polynomial', KerSpDegree=2, KerIntKnotX=(), KerIntKnotY=(),
               BkgSpType='Polynomial', BkgSpDegree=2, BkgIntKnotX=(), BkgIntKnotY=()):
    """Tabulated 1-D factors and (x-factor, y-factor) pair lists in the reference's REF_ij / REF_pq order
    (SFFTSubtract.py:627-636 of the dev version)."""
    DK, DB = int(KerSpDegree), int(BkgSpDegree)
    cx = (np.arange(N0, dtype=np.float64) + 1.0) / N0
    cy = (np.arange(N1, dtype=np.float64) + 1.0) / N1
    if KerSpType == 'Polynomial':
        kbx = np.stack([np.power(cx, i) for i in range(DK + 1)])
        kby = np.stack([np.power(cy, j) for j in range(DK + 1)])
        ker_pairs = [(i, j) for i in range(DK + 1) for j in range(DK + 1 - i)]
    else:
        kbx = create_bspline_basis(N0, list(KerIntKnotX), DK)
        kby = create_bspline_basis(N1, list(KerIntKnotY), DK)
        ker_pairs = [(i, j) for i in range(kbx.shape[0]) for j in range(kby.shape[0])]
    if BkgSpType == 'Polynomial':
        tbx = np.stack([np.power(cx, p) for p in range(DB + 1)])
        tby = np.stack([np.power(cy, q) for q in range(DB + 1)])
        bkg_pairs = [(p, q) for p in range(DB + 1) for q in range(DB + 1 - p)]
    else:
        tbx = create_bspline_basis(N0, list(BkgIntKnotX), DB)
        tby = create_bspline_basis(N1, list(BkgIntKnotY), DB)
        bkg_pairs = [(p, q) for p in range(tbx.shape[0]) for q in range(tby.shape[0])]
    return dict(kbx=kbx, kby=kby, ker_pairs=np.array(ker_pairs, dtype=np.int32),
                tbx=tbx, tby=tby, bkg_pairs=np.array(bkg_pairs, dtype=np.int32),
                KerSpType=KerSpType, BkgSpType=BkgSpType)


def SSC(NX, NY, KerHW, basis, ConstPhotRatio=True):
    """Parameter dictionary (SFFTConfigure.py:1077-1150 of the dev version)."""
    N0, N1, w = int(NX), int(NY), int(KerHW)
    L = 2 * w + 1
    Fab = L * L
    Fij, Fpq = len(basis['ker_pairs']), len(basis['bkg_pairs'])
    return dict(N0=N0, N1=N1, w0=w, w1=w, L0=L, L1=L, Fab=Fab, Fij=Fij, Fpq=Fpq, Fijab=Fij * Fab,
                NEQ=Fij * Fab + Fpq, NEQt=Fij * Fab + Fpq - Fij + 1, SCALE=np.float64(1 / (N0 * N1)),
                SCALE_L=np.float64(N0 * N1), ConstPhotRatio=ConstPhotRatio, KerSpType=basis['KerSpType'])


def _planes(PixA_I, basis):
    Iij = np.stack([PixA_I * (basis['kbx'][i][:, None] * basis['kby'][j][None, :]) for i, j in basis['ker_pairs']])
    Tpq = np.stack([basis['tbx'][p][:, None] * basis['tby'][q][None, :] for p, q in basis['bkg_pairs']])
    return Iij, Tpq


def establish_system(PixA_I, PixA_J, p, basis, workers=1):
    """LHMAT[NEQ,NEQ], RHb[NEQ] before TweakLS (SFFTSubtract.py:728-868 of the dev version; fill rules identical to
    sfft/sfftcore, SFFTConfigure.py:1291-1610 there)."""
    N0, N1 = p['N0'], p['N1']
    Fij, Fpq, Fab, Fijab, NEQ = p['Fij'], p['Fpq'], p['Fab'], p['Fijab'], p['NEQ']
    SCALE, SCALE_L = p['SCALE'], p['SCALE_L']
    w0, w1, L0, L1 = p['w0'], p['w1'], p['L0'], p['L1']
    Iij, Tpq = _planes(PixA_I, basis)
    FJ = SCALE * _fft2(PixA_J.astype(np.complex128), workers)
    FI = np.stack([SCALE * _fft2(Iij[k].astype(np.complex128), workers) for k in range(Fij)])
    FT = np.stack([SCALE * _fft2(Tpq[k].astype(np.complex128), workers) for k in range(Fpq)])
    del Iij, Tpq
    CFJ, CFI, CFT = np.conj(FJ), np.conj(FI), np.conj(FT)
    ab = np.array([(a - w0, b - w1) for a in range(L0) for b in range(L1)])
    a_, b_ = ab[:, 0], ab[:, 1]
    cen = (a_ == 0) & (b_ == 0)
    MODa, MODb = np.mod(a_, N0), np.mod(b_, N1)
    MOD_a, MOD_b = np.mod(-a_, N0), np.mod(-b_, N1)
    MODda, MODdb = np.mod(a_[:, None] - a_[None, :], N0), np.mod(b_[:, None] - b_[None, :], N1)
    LHMAT = np.empty((NEQ, NEQ))
    RHb = np.empty(NEQ)
    for i8 in range(Fij):
        for ij in range(Fij):
            Pre = (SCALE * _fft2(FI[i8] * CFI[ij], workers)).real * SCALE
            P00, Prow, Pcol = Pre[0, 0], Pre[MODa, MODb], Pre[MOD_a, MOD_b]
            blk = -Prow[:, None] - Pcol[None, :] + Pre[MODda, MODdb] + P00
            blk[cen, :] = (Pcol - P00)[None, :]
            blk[:, cen] = (Prow - P00)[:, None]
            blk[np.ix_(cen, cen)] = P00
            LHMAT[i8 * Fab:(i8 + 1) * Fab, ij * Fab:(ij + 1) * Fab] = blk
        for pq in range(Fpq):
            Pre = (SCALE * _fft2(FI[i8] * CFT[pq], workers)).real
            col = Pre[MODa, MODb] - Pre[0, 0]
            col[cen] = Pre[0, 0]
            LHMAT[i8 * Fab:(i8 + 1) * Fab, Fijab + pq] = col
        Pre = (SCALE * _fft2(CFJ * FI[i8], workers)).real
        v = Pre[MODa, MODb] - Pre[0, 0]
        v[cen] = Pre[0, 0]
        RHb[i8 * Fab:(i8 + 1) * Fab] = v
    for p8 in range(Fpq):
        for ij in range(Fij):
            Pre = (SCALE * _fft2(FT[p8] * CFI[ij], workers)).real
            row = Pre[MOD_a, MOD_b] - Pre[0, 0]
            row[cen] = Pre[0, 0]
            LHMAT[Fijab + p8, ij * Fab:(ij + 1) * Fab] = row
        for pq in range(Fpq):
            LHMAT[Fijab + p8, Fijab + pq] = ((SCALE * _fft2(FT[p8] * CFT[pq], workers)).real * SCALE_L)[0, 0]
        RHb[Fijab + p8] = ((SCALE * _fft2(CFJ * FT[p8], workers)).real * SCALE_L)[0, 0]
    return LHMAT, RHb


def solve_system(LHMAT, RHb, p):
    """TweakLS + gesv + Restore_Solution (SFFTSubtract.py:870-902 of the dev version; BSplineSFFT.py:3702-3768)."""
    NEQ, Fijab, Fab = p['NEQ'], p['Fijab'], p['Fab']
    if not p['ConstPhotRatio']:
        return np.linalg.solve(LHMAT, RHb).astype(np.float64)
    ij00 = np.arange(p['w0'] * p['L1'] + p['w1'], Fijab, Fab)
    PresIDX = np.setdiff1d(np.arange(NEQ), ij00[1:], assume_unique=True)
    if p['KerSpType'] == 'Polynomial':
        A = LHMAT[np.ix_(PresIDX, PresIDX)]
        b = RHb[PresIDX]
        Solution = np.zeros(NEQ)
        Solution[PresIDX] = np.linalg.solve(A, b)
        return Solution
    # B-spline: the ij00 unknowns are one unknown -- sum their rows and columns into the first of them
    P = np.zeros((NEQ, len(PresIDX)))
    P[PresIDX, np.arange(len(PresIDX))] = 1.0
    key = int(np.where(PresIDX == ij00[0])[0][0])
    P[ij00[1:], key] = 1.0
    xt = np.linalg.solve(P.T @ LHMAT @ P, P.T @ RHb)
    return (P @ xt).astype(np.float64)


def subtract(PixA_I, PixA_J, Solution, p, basis, workers=1):
    """SFFTSubtract.py:925-968 of the dev version (Construct_FDIFF as a matrix product, see sfft_oracle.construct_fdiff)."""
    N0, N1, w0, w1, L0, L1 = p['N0'], p['N1'], p['w0'], p['w1'], p['L0'], p['L1']
    Fij, Fpq, Fab, Fijab, SCALE, SCALE_L = p['Fij'], p['Fpq'], p['Fab'], p['Fijab'], p['SCALE'], p['SCALE_L']
    Iij, Tpq = _planes(PixA_I, basis)
    FJ = SCALE * _fft2(PixA_J.astype(np.complex128), workers)
    Wl = np.exp((-2j * np.pi / N0) * np.arange(N0))
    Wm = np.exp((-2j * np.pi / N1) * np.arange(N1))
    Wla = np.stack([Wl ** a for a in range(-w0, w0 + 1)])
    Wmb = np.stack([Wm ** b for b in range(-w1, w1 + 1)])
    a_ijab = Solution[:Fijab].astype(np.complex128)
    b_pq = Solution[Fijab:].astype(np.complex128)
    cen = w0 * L1 + w1
    PVAL = np.zeros((N0, N1), dtype=np.complex128)
    for ij in range(Fij):
        FIij = SCALE * _fft2(Iij[ij].astype(np.complex128), workers)
        A = a_ijab[ij * Fab:(ij + 1) * Fab].reshape(L0, L1)
        S_off = A.sum() - A.reshape(-1)[cen]
        PVAL += FIij * (SCALE * ((Wla.T @ A) @ Wmb - S_off))
    for pq in range(Fpq):
        PVAL += b_pq[pq] * (SCALE * _fft2(Tpq[pq].astype(np.complex128), workers))
    return np.ascontiguousarray((SCALE_L * _ifft2(FJ - PVAL, workers)).real)


def ESS(PixA_I, PixA_J, p, basis, SFFTSolution=None, Subtract=False, workers=1):
    assert PixA_I.shape == (p['N0'], p['N1']) and PixA_J.shape == (p['N0'], p['N1'])
    PixA_I = np.ascontiguousarray(PixA_I, np.float64)
    PixA_J = np.ascontiguousarray(PixA_J, np.float64)
    if SFFTSolution is None:
        LHMAT, RHb = establish_system(PixA_I, PixA_J, p, basis, workers)
        Solution = solve_system(LHMAT, RHb, p)
    else:
        Solution = np.asarray(SFFTSolution, dtype=np.float64)
    DIFF = subtract(PixA_I, PixA_J, Solution, p, basis, workers) if Subtract else None
    return Solution, DIFF


def GSS(PixA_I, PixA_J, PixA_mI, PixA_mJ, p, basis, workers=1):
    Solution = ESS(PixA_mI, PixA_mJ, p, basis, None, False, workers)[0]
    DIFF = ESS(PixA_I, PixA_J, p, basis, Solution, True, workers)[1]
    return Solution, DIFF
