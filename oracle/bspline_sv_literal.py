"""LITERAL numpy transcription of the CuPy-only parts of sfft/BSplineSFFT.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

oracle/bspline_sv_oracle.py restates SCALING_MODE 'SEPARATE-VARYING' and REGULARIZE_KERNEL in vectorised form.  The reference
cannot run that code in the build container (CuPy only), so nothing the reference produced pins it.  This file NARROWS
that gap without closing it: it is a second, independent transcription that follows the reference's CUDA kernels and host
loops line by line -- one Python loop nest per `kmain`, the same index tables, the same `cIdx` loops, the same sequence of
scalings -- so that a reading mistake would have to be made twice, in two differently shaped programs, to go unnoticed.
tests/test_oracle_sv.py requires the two to agree to 1e-13 (relative to the block maximum) on small cases.

Parity status: the vectorised oracle this file cross-checks is pinned since round 3 through the reference's NIRCam golden
(oracle/nircam_chain.py); this file itself is not run on it; "two transcriptions agree" is evidence, not
a golden vector.  Only tests/ may import this module.

Kernels transcribed (sfft/BSplineSFFT.py, v1.7.3):
    ScaSpatial                     :334-397      HadProd_OMG11/01/10/00        :1354-1471     FillLS_OMG   :1473-1558
    HadProd_GAM1/GAM0              :1560-1616    FillLS_GAM                    :1618-1672
    HadProd_PSI1/PSI0              :1674-1730    FillLS_PSI                    :1732-1786
    HadProd_PHI / FillLS_PHI       :1788-1851    HadProd_THE1/THE0, FillLS_THE :1853-1952     HadProd_DEL / FillLS_DEL :1954-2004
    fill_lapmat_nondiagonal        :2009-2046    fill_iregmat                  :2048-2088     fill_regmat  :2090-2168
    TweakLS / Restore_Solution     :2171-2342    Construct_FDIFF (varying)     :2429-2527
host sequence: ESS :2750-2830 (index tables), :3296-3570 (Greek loops), :3572-3700 (regularisation), :3702-3790 (tweak, solve).
"""
import numpy as np
from scipy import signal


def _mod(v, N):
    """`tmp = fmod(float(v), float(N)); if (tmp < 0.0) tmp += float(N); int M = tmp;`"""
    tmp = np.fmod(np.float32(v), np.float32(N))
    if tmp < 0.0:
        tmp += np.float32(N)
    return int(tmp)


def index_tables(Fij, Fpq, w0, w1):
    """ESS :2761-2800."""
    L0, L1 = 2 * w0 + 1, 2 * w1 + 1
    Fab = L0 * L1
    T = {}
    T['REF_ab'] = np.array([(a_pos - w0, b_pos - w1) for a_pos in range(L0) for b_pos in range(L1)]).astype(np.int32)
    T['SREF_iji0j0'] = np.array([(ij, i0j0) for ij in range(Fij) for i0j0 in range(Fij)]).astype(np.int32)
    T['SREF_pqp0q0'] = np.array([(pq, p0q0) for pq in range(Fpq) for p0q0 in range(Fpq)]).astype(np.int32)
    T['SREF_ijpq'] = np.array([(ij, pq) for ij in range(Fij) for pq in range(Fpq)]).astype(np.int32)
    T['SREF_pqij'] = np.array([(pq, ij) for pq in range(Fpq) for ij in range(Fij)]).astype(np.int32)
    T['SREF_ijab'] = np.array([(ij, ab) for ij in range(Fij) for ab in range(Fab)]).astype(np.int32)
    T['ij00'] = np.arange(w0 * L1 + w1, Fij * Fab, Fab).astype(np.int32)
    return T


def sca_spatial(PixA_I, sbx, sby, ScaREF_ij):
    """ScaSpatial :334-397: ScaSPixA_Iij[ij] = I * (x factor i) * (y factor j); a place-holder (-1, -1) gives a zero plane.
    (The reference's B-spline variant indexes its basis tables with -1 for a place-holder, which reads out of bounds; its
    polynomial variant writes zeros, and zeros are what the rest of the code assumes.)"""
    N0, N1 = PixA_I.shape
    out = np.zeros((len(ScaREF_ij), N0, N1))
    for ij in range(len(ScaREF_ij)):
        i, j = ScaREF_ij[ij]
        if i >= 0 and j >= 0:
            for ROW in range(N0):
                for COL in range(N1):
                    out[ij][ROW][COL] = PixA_I[ROW][COL] * (sbx[i][ROW] * sby[j][COL])
    return out


def establish_system(PixA_I, PixA_J, N0, N1, w0, w1, kbx, kby, ker_pairs, sbx, sby, ScaREF_ij, tbx, tby, bkg_pairs):
    """ESS :3296-3570, SEPARATE-VARYING: returns (LHMAT[NEQ, NEQ], RHb[NEQ]) before regularisation and TweakLS."""
    L0, L1 = 2 * w0 + 1, 2 * w1 + 1
    Fab = L0 * L1
    Fij, Fpq = len(ker_pairs), len(bkg_pairs)
    Fijab, NEQ = Fij * Fab, Fij * Fab + Fpq
    FOMG, FGAM, FPSI, FPHI, FTHE, FDEL = Fij * Fij, Fij * Fpq, Fpq * Fij, Fpq * Fpq, Fij, Fpq
    SCALE = np.float64(1 / (N0 * N1))
    SCALE_L = np.float64(1 / SCALE)
    T = index_tables(Fij, Fpq, w0, w1)
    REF_ab, SREF_ijab = T['REF_ab'], T['SREF_ijab']

    # spatial planes and their (scaled) transforms, :2832-3030
    SPixA_Iij = np.array([PixA_I * (kbx[i][:, None] * kby[j][None, :]) for i, j in ker_pairs])
    ScaSPixA_Iij = sca_spatial(PixA_I, sbx, sby, ScaREF_ij)
    SPixA_Tpq = np.array([tbx[p][:, None] * tby[q][None, :] for p, q in bkg_pairs])
    PixA_FJ = np.fft.fft2(PixA_J.astype(np.complex128)) * SCALE
    SPixA_FIij = np.array([np.fft.fft2(x.astype(np.complex128)) * SCALE for x in SPixA_Iij])
    ScaSPixA_FIij = np.array([np.fft.fft2(x.astype(np.complex128)) * SCALE for x in ScaSPixA_Iij])
    SPixA_FTpq = np.array([np.fft.fft2(x.astype(np.complex128)) * SCALE for x in SPixA_Tpq])
    PixA_CFJ, SPixA_CFIij = np.conj(PixA_FJ), np.conj(SPixA_FIij)
    ScaSPixA_CFIij, SPixA_CFTpq = np.conj(ScaSPixA_FIij), np.conj(SPixA_FTpq)

    LHMAT = np.empty((NEQ, NEQ), dtype=np.float64)
    RHb = np.empty(NEQ, dtype=np.float64)

    def pre(Hp, post):
        """`Hp = fft2(Hp); Hp *= SCALE; Pre = Hp.real; Pre *= post`"""
        Hp = np.fft.fft2(Hp)
        Hp *= SCALE
        Pre = np.empty((N0, N1), dtype=np.float64)
        Pre[:, :] = Hp.real
        if post is not None:
            Pre[:, :] *= post
        return Pre

    # OMEGA :3298-3385
    for cIdx in range(FOMG):
        i8j8, ij = T['SREF_iji0j0'][cIdx]
        cPreOMG11 = pre(SPixA_FIij[i8j8] * SPixA_CFIij[ij], SCALE)               # HadProd_OMG11 :1354-1381
        cPreOMG01 = pre(ScaSPixA_FIij[i8j8] * SPixA_CFIij[ij], SCALE)            # HadProd_OMG01 :1383-1411
        cPreOMG10 = pre(SPixA_FIij[i8j8] * ScaSPixA_CFIij[ij], SCALE)            # HadProd_OMG10 :1413-1441
        cPreOMG00 = pre(ScaSPixA_FIij[i8j8] * ScaSPixA_CFIij[ij], SCALE)         # HadProd_OMG00 :1443-1471
        for ROW in range(Fijab):                                                  # FillLS_OMG :1473-1558
            for COL in range(Fijab):
                i8j8_, a8b8 = SREF_ijab[ROW]
                ij_, ab = SREF_ijab[COL]
                a8, b8 = REF_ab[a8b8]
                a, b = REF_ab[ab]
                idx = i8j8_ * Fij + ij_
                if idx == cIdx:
                    MODa8, MODb8 = _mod(a8, N0), _mod(b8, N1)
                    MOD_a, MOD_b = _mod(-a, N0), _mod(-b, N1)
                    MODa8_a, MODb8_b = _mod(a8 - a, N0), _mod(b8 - b, N1)
                    if (a8 != 0 or b8 != 0) and (a != 0 or b != 0):
                        LHMAT[ROW][COL] = - cPreOMG11[MODa8][MODb8] - cPreOMG11[MOD_a][MOD_b] \
                            + cPreOMG11[MODa8_a][MODb8_b] + cPreOMG11[0][0]
                    if (a8 == 0 and b8 == 0) and (a != 0 or b != 0):
                        LHMAT[ROW][COL] = cPreOMG01[MOD_a][MOD_b] - cPreOMG01[0][0]
                    if (a8 != 0 or b8 != 0) and (a == 0 and b == 0):
                        LHMAT[ROW][COL] = cPreOMG10[MODa8][MODb8] - cPreOMG10[0][0]
                    if (a8 == 0 and b8 == 0) and (a == 0 and b == 0):
                        LHMAT[ROW][COL] = cPreOMG00[0][0]

    # GAMMA :3389-3425
    for cIdx in range(FGAM):
        i8j8, pq = T['SREF_ijpq'][cIdx]
        cPreGAM1 = pre(SPixA_FIij[i8j8] * SPixA_CFTpq[pq], None)                 # HadProd_GAM1 :1560-1587
        cPreGAM0 = pre(ScaSPixA_FIij[i8j8] * SPixA_CFTpq[pq], None)              # HadProd_GAM0 :1589-1616
        for ROW in range(Fijab):                                                  # FillLS_GAM :1618-1672
            for COL in range(Fpq):
                i8j8_, a8b8 = SREF_ijab[ROW]
                a8, b8 = REF_ab[a8b8]
                idx = i8j8_ * Fpq + COL
                cCOL = Fijab + COL
                if idx == cIdx:
                    MODa8, MODb8 = _mod(a8, N0), _mod(b8, N1)
                    if a8 != 0 or b8 != 0:
                        LHMAT[ROW][cCOL] = cPreGAM1[MODa8][MODb8] - cPreGAM1[0][0]
                    if a8 == 0 and b8 == 0:
                        LHMAT[ROW][cCOL] = cPreGAM0[0][0]

    # PSI :3429-3460
    for cIdx in range(FPSI):
        p8q8, ij = T['SREF_pqij'][cIdx]
        cPrePSI1 = pre(SPixA_CFIij[ij] * SPixA_FTpq[p8q8], None)                 # HadProd_PSI1 :1674-1701
        cPrePSI0 = pre(ScaSPixA_CFIij[ij] * SPixA_FTpq[p8q8], None)              # HadProd_PSI0 :1703-1730
        for ROW in range(Fpq):                                                    # FillLS_PSI :1732-1786
            for COL in range(Fijab):
                cROW = Fijab + ROW
                ij_, ab = SREF_ijab[COL]
                a, b = REF_ab[ab]
                idx = ROW * Fij + ij_
                if idx == cIdx:
                    MOD_a, MOD_b = _mod(-a, N0), _mod(-b, N1)
                    if a != 0 or b != 0:
                        LHMAT[cROW][COL] = cPrePSI1[MOD_a][MOD_b] - cPrePSI1[0][0]
                    if a == 0 and b == 0:
                        LHMAT[cROW][COL] = cPrePSI0[0][0]

    # PHI :3464-3482
    for cIdx in range(FPHI):
        p8q8, pq = T['SREF_pqp0q0'][cIdx]
        cPrePHI = pre(SPixA_FTpq[p8q8] * SPixA_CFTpq[pq], SCALE_L)                # HadProd_PHI :1788-1817
        for ROW in range(Fpq):                                                    # FillLS_PHI :1819-1851
            for COL in range(Fpq):
                if ROW * Fpq + COL == cIdx:
                    LHMAT[Fijab + ROW][Fijab + COL] = cPrePHI[0][0]

    # THETA, DELTA :3486-3540
    PreTHE1 = np.array([pre(SPixA_FIij[k] * PixA_CFJ, None) for k in range(FTHE)])       # HadProd_THE1 :1853-1877
    PreTHE0 = np.array([pre(ScaSPixA_FIij[k] * PixA_CFJ, None) for k in range(FTHE)])    # HadProd_THE0 :1879-1903
    PreDEL = np.array([pre(SPixA_FTpq[k] * PixA_CFJ, SCALE_L) for k in range(FDEL)])     # HadProd_DEL :1954-1978
    for ROW in range(Fijab):                                                      # FillLS_THE :1905-1952
        i8j8, a8b8 = SREF_ijab[ROW]
        a8, b8 = REF_ab[a8b8]
        MODa8, MODb8 = _mod(a8, N0), _mod(b8, N1)
        if a8 != 0 or b8 != 0:
            RHb[ROW] = PreTHE1[i8j8][MODa8][MODb8] - PreTHE1[i8j8][0][0]
        if a8 == 0 and b8 == 0:
            RHb[ROW] = PreTHE0[i8j8][0][0]
    for ROW in range(Fpq):                                                        # FillLS_DEL :1980-2004
        RHb[Fijab + ROW] = PreDEL[ROW][0][0]
    return LHMAT, RHb


def regularization_matrix(N0, N1, w0, w1, Fij, Fpq, SSTMAT, CSSTMAT=None, DSSTMAT=None, IGNORE_LAPLACIAN_KERCENT=True):
    """:3637-3695 with the kernels fill_lapmat_nondiagonal :2009-2046, fill_iregmat :2048-2088, fill_regmat :2090-2168.
    CSSTMAT / DSSTMAT given -> the SEPARATE-VARYING form of fill_regmat.  Returns REGMAT[NEQ, NEQ] (to be added times lambda)."""
    L0, L1 = 2 * w0 + 1, 2 * w1 + 1
    Fab = L0 * L1
    Fijab, NEQ = Fij * Fab, Fij * Fab + Fpq
    SCALE2 = np.float64(1 / (N0 * N1)) ** 2
    LAPMAT = np.zeros((Fab, Fab)).astype(np.int32)
    RR, CC = np.mgrid[0: L0, 0: L1]
    RRF, CCF = RR.flatten().astype(np.int32), CC.flatten().astype(np.int32)
    AdCOUNT = signal.correlate2d(np.ones((L0, L1)), np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]]), mode='same', boundary='fill',
                                 fillvalue=0).astype(np.int32)
    KIDX = np.arange(Fab)
    LAPMAT[KIDX, KIDX] = AdCOUNT.flatten()[KIDX]
    for ROW in range(Fab):                                                        # fill_lapmat_nondiagonal
        for COL in range(Fab):
            if ROW != COL:
                r1, c1, r2, c2 = RRF[ROW], CCF[ROW], RRF[COL], CCF[COL]
                if r2 == r1 - 1 and c2 == c1:
                    LAPMAT[ROW][COL] = -1
                if r2 == r1 + 1 and c2 == c1:
                    LAPMAT[ROW][COL] = -1
                if r2 == r1 and c2 == c1 - 1:
                    LAPMAT[ROW][COL] = -1
                if r2 == r1 and c2 == c1 + 1:
                    LAPMAT[ROW][COL] = -1
    if IGNORE_LAPLACIAN_KERCENT:
        LAPMAT[(w0 - 1) * L1 + w1, :] = 0.0
        LAPMAT[w0 * L1 + w1 - 1, :] = 0.0
        LAPMAT[w0 * L1 + w1, :] = 0.0
        LAPMAT[w0 * L1 + w1 + 1, :] = 0.0
        LAPMAT[(w0 + 1) * L1 + w1, :] = 0.0
    LTLMAT = np.matmul(LAPMAT.T, LAPMAT)
    c0 = w0 * L1 + w1
    iREGMAT = np.zeros((Fab, Fab), dtype=np.int32)
    for ROW in range(Fab):                                                        # fill_iregmat
        for COL in range(Fab):
            if ROW != c0 and COL != c0:
                iREGMAT[ROW][COL] = LTLMAT[ROW][COL] + LTLMAT[COL][ROW] - LTLMAT[c0][ROW] - LTLMAT[c0][COL] \
                    - LTLMAT[ROW][c0] - LTLMAT[COL][c0] + 2 * LTLMAT[c0][c0]
            if ROW != c0 and COL == c0:
                iREGMAT[ROW][COL] = LTLMAT[ROW][c0] + LTLMAT[c0][ROW] - 2 * LTLMAT[c0][c0]
            if ROW == c0 and COL != c0:
                iREGMAT[ROW][COL] = LTLMAT[COL][c0] + LTLMAT[c0][COL] - 2 * LTLMAT[c0][c0]
            if ROW == c0 and COL == c0:
                iREGMAT[ROW][COL] = 2 * LTLMAT[c0][c0]
    REGMAT = np.zeros((NEQ, NEQ), dtype=np.float64)
    for ROW in range(Fijab):                                                      # fill_regmat
        for COL in range(Fijab):
            k, c = ROW // Fab, ROW % Fab
            k8, c8 = COL // Fab, COL % Fab
            if CSSTMAT is None:
                REGMAT[ROW][COL] = SCALE2 * SSTMAT[k][k8] * iREGMAT[c][c8]
            else:
                if c != c0 and c8 != c0:
                    REGMAT[ROW][COL] = SCALE2 * SSTMAT[k][k8] * iREGMAT[c][c8]
                if c != c0 and c8 == c0:
                    REGMAT[ROW][COL] = SCALE2 * CSSTMAT[k][k8] * iREGMAT[c][c8]
                if c == c0 and c8 != c0:
                    REGMAT[ROW][COL] = SCALE2 * CSSTMAT[k8][k] * iREGMAT[c][c8]
                if c == c0 and c8 == c0:
                    REGMAT[ROW][COL] = SCALE2 * DSSTMAT[k][k8] * iREGMAT[c][c8]
    return REGMAT, iREGMAT


def spatial_grams(SPMAT, ScaSPMAT, Fij, WEIGHT_REGULARIZE=None):
    """:3583-3635: SSTMAT, CSSTMAT, DSSTMAT from the basis values at the regularisation points (place-holder rows of zeros)."""
    NREG = SPMAT.shape[1]
    if ScaSPMAT is not None and ScaSPMAT.shape[0] < Fij:
        ScaSPMAT = np.concatenate((ScaSPMAT, np.zeros((Fij - ScaSPMAT.shape[0], NREG), dtype=np.float64)), axis=0)
    if WEIGHT_REGULARIZE is None:
        SST = np.matmul(SPMAT, SPMAT.T) / NREG
        CSST = None if ScaSPMAT is None else np.matmul(SPMAT, ScaSPMAT.T) / NREG
        DSST = None if ScaSPMAT is None else np.matmul(ScaSPMAT, ScaSPMAT.T) / NREG
    else:
        WSPMAT = np.diag(np.asarray(WEIGHT_REGULARIZE, dtype=np.float64))
        WSPMAT /= np.sum(WEIGHT_REGULARIZE)
        SST = np.matmul(np.matmul(SPMAT, WSPMAT), SPMAT.T)
        CSST = None if ScaSPMAT is None else np.matmul(np.matmul(SPMAT, WSPMAT), ScaSPMAT.T)
        DSST = None if ScaSPMAT is None else np.matmul(np.matmul(ScaSPMAT, WSPMAT), ScaSPMAT.T)
    return SST, CSST, DSST


def tweak_solve_restore(LHMAT, RHb, Fij, Fpq, w0, w1, SCALING_MODE, KerSpType, ScaFij=None):
    """:3702-3790 with the kernels TweakLS :2171-2272 / :2293-2318 and Restore_Solution :2274-2291 / :2320-2342."""
    L1 = 2 * w1 + 1
    Fab = (2 * w0 + 1) * L1
    Fijab, NEQ = Fij * Fab, Fij * Fab + Fpq
    ij00 = np.arange(w0 * L1 + w1, Fijab, Fab).astype(np.int32)
    if SCALING_MODE == 'ENTANGLED' or (SCALING_MODE == 'SEPARATE-VARYING' and ScaFij == Fij):
        return np.linalg.solve(LHMAT, RHb)
    if SCALING_MODE == 'SEPARATE-CONSTANT':
        PresIDX = np.setdiff1d(np.arange(NEQ), ij00[1:], assume_unique=True).astype(np.int32)
    else:
        PresIDX = np.setdiff1d(np.arange(NEQ), ij00[ScaFij:], assume_unique=True).astype(np.int32)
    NEQt = len(PresIDX)
    assert np.all(PresIDX[:-1] < PresIDX[1:]) and PresIDX[ij00[0]] == ij00[0]
    LHt = np.empty((NEQt, NEQt), dtype=np.float64)
    RHt = np.empty(NEQt, dtype=np.float64)
    if SCALING_MODE == 'SEPARATE-CONSTANT' and KerSpType == 'B-Spline':
        keyIdx = ij00[0]
        for ROW in range(NEQt):
            for COL in range(NEQt):
                if ROW == keyIdx and COL != keyIdx:
                    cum1 = 0.0
                    for ij in range(Fij):
                        cum1 += LHMAT[ij00[ij]][PresIDX[COL]]
                    LHt[ROW][COL] = cum1
                if ROW != keyIdx and COL == keyIdx:
                    cum2 = 0.0
                    for ij in range(Fij):
                        cum2 += LHMAT[PresIDX[ROW]][ij00[ij]]
                    LHt[ROW][COL] = cum2
                if ROW == keyIdx and COL == keyIdx:
                    cum3 = 0.0
                    for ij in range(Fij):
                        for i8j8 in range(Fij):
                            cum3 += LHMAT[ij00[ij]][ij00[i8j8]]
                    LHt[ROW][COL] = cum3
                if ROW != keyIdx and COL != keyIdx:
                    LHt[ROW][COL] = LHMAT[PresIDX[ROW]][PresIDX[COL]]
            if ROW == keyIdx:
                cum4 = 0.0
                for ij in range(Fij):
                    cum4 += RHb[ij00[ij]]
                RHt[ROW] = cum4
            else:
                RHt[ROW] = RHb[PresIDX[ROW]]
    else:
        for ROW in range(NEQt):
            for COL in range(NEQt):
                LHt[ROW][COL] = LHMAT[PresIDX[ROW]][PresIDX[COL]]
            RHt[ROW] = RHb[PresIDX[ROW]]
    St = np.linalg.solve(LHt, RHt)
    Solution = np.zeros(NEQ, dtype=np.float64)
    if SCALING_MODE == 'SEPARATE-CONSTANT' and KerSpType == 'B-Spline':
        Solution[ij00[1:]] = St[ij00[0]]
    for ROW in range(NEQt):                                                       # Restore_Solution
        Solution[PresIDX[ROW]] = St[ROW]
    return Solution


def construct_diff(PixA_I, PixA_J, Solution, N0, N1, w0, w1, kbx, kby, ker_pairs, sbx, sby, ScaREF_ij, tbx, tby, bkg_pairs):
    """Construct_FDIFF for SEPARATE-VARYING :2429-2527 and the inverse transform (ESS :3820-3860): per pixel, over ab then ij; the
    centre element of the kernel uses the SCALING planes."""
    L0, L1 = 2 * w0 + 1, 2 * w1 + 1
    Fab = L0 * L1
    Fij, Fpq = len(ker_pairs), len(bkg_pairs)
    Fijab = Fij * Fab
    SCALE = np.float64(1 / (N0 * N1))
    SCALE_L = np.float64(1 / SCALE)
    T = index_tables(Fij, Fpq, w0, w1)
    REF_ab = T['REF_ab']
    SPixA_FIij = np.array([np.fft.fft2((PixA_I * (kbx[i][:, None] * kby[j][None, :])).astype(np.complex128)) * SCALE for i, j in ker_pairs])
    ScaSPixA_FIij = np.array([np.fft.fft2(x.astype(np.complex128)) * SCALE for x in sca_spatial(PixA_I, sbx, sby, ScaREF_ij)])
    SPixA_FTpq = np.array([np.fft.fft2((tbx[p][:, None] * tby[q][None, :]).astype(np.complex128)) * SCALE for p, q in bkg_pairs])
    PixA_FJ = np.fft.fft2(PixA_J.astype(np.complex128)) * SCALE
    PixA_X = np.arange(N0)[:, None] * np.ones((1, N1))
    PixA_Y = np.ones((N0, 1)) * np.arange(N1)[None, :]
    Wl = np.exp((-2j * np.pi / N0) * PixA_X.astype(np.float64))
    Wm = np.exp((-2j * np.pi / N1) * PixA_Y.astype(np.float64))
    Kab_Wla = np.array([Wl ** a for a in range(-w0, w0 + 1)])
    Kab_Wmb = np.array([Wm ** b for b in range(-w1, w1 + 1)])
    a_ijab = Solution[:Fijab].astype(np.complex128)
    b_pq = Solution[Fijab:].astype(np.complex128)
    SCA = np.complex128(SCALE)
    PixA_FDIFF = np.empty((N0, N1), dtype=np.complex128)
    for ROW in range(N0):
        for COL in range(N1):
            PVAL = 0.0 + 0.0j
            for ab in range(Fab):
                a, b = REF_ab[ab]
                if a == 0 and b == 0:
                    PVAL_FKab = SCA
                    for ij in range(Fij):
                        PVAL = PVAL + (a_ijab[ij * Fab + ab] * ScaSPixA_FIij[ij][ROW][COL]) * PVAL_FKab
                if a != 0 or b != 0:
                    PVAL_FKab = SCA * (Kab_Wla[w0 + a][ROW][COL] * Kab_Wmb[w1 + b][ROW][COL] - 1.0)
                    for ij in range(Fij):
                        PVAL = PVAL + (a_ijab[ij * Fab + ab] * SPixA_FIij[ij][ROW][COL]) * PVAL_FKab
            for pq in range(Fpq):
                PVAL = PVAL + b_pq[pq] * SPixA_FTpq[pq][ROW][COL]
            PixA_FDIFF[ROW][COL] = PixA_FJ[ROW][COL] - PVAL
    return (SCALE_L * np.fft.ifft2(PixA_FDIFF)).real


def convolve2d_same_fill0(img, ker):
    """scipy.signal.convolve2d(img, ker, mode='same', boundary='fill', fillvalue=0) written out as its definition:
    out[x, y] = sum_{u, v} ker[u, v] * img[x + c0 - u, y + c1 - v], zero outside the image, (c0, c1) = (L0 - 1) // 2, (L1 - 1) // 2."""
    N0, N1 = img.shape
    L0, L1 = ker.shape
    c0, c1 = (L0 - 1) // 2, (L1 - 1) // 2
    out = np.zeros((N0, N1))
    for x in range(N0):
        for y in range(N1):
            acc = 0.0
            for u in range(L0):
                for v in range(L1):
                    xx, yy = x + c0 - u, y + c1 - v
                    if 0 <= xx < N0 and 0 <= yy < N1:
                        acc += ker[u, v] * img[xx, yy]
            out[x, y] = acc
    return out
