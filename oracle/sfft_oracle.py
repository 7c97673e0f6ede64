"""CPU oracle for the SFFT subtraction hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a numpy/scipy restatement of the reference's Numpy backend
(thomasvrussell/sfft v1.7.3):

    sfft/sfftcore/SFFTConfigure.py:817-1367   SingleSFFTConfigure_Numpy.SSCN (the 17 njit functions)
    sfft/sfftcore/SFFTSubtract.py:477-821     ElementalSFFTSubtract_Numpy.ESSN
    sfft/sfftcore/SFFTSubtract.py:839-923     GeneralSFFTSubtract.GSS
    sfft/CustomizedPacket.py:114-188          packet-level NaN / ForceConv / sign handling

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
nothing under sfft_amd/ does.  The shipped path is the HIP library and it fails
loudly when that library is missing.

Parity status: PINNED.  tests/golden/*.npz hold LHMAT / RHb / Solution / DIFF produced
by importing the reference's own sfftcore modules in the build container
(tests/golden/make_golden.py); tests/test_oracle_golden.py checks this file against
every one of them.

The arithmetic follows the reference step by step (same planes, same scalings, same
fill rules, LAPACK gesv solve); loops that the reference runs under numba.prange are
vectorised with numpy indexing.  Two re-groupings are used for speed and are checked
against literal transcriptions (`*_literal`) in the tests:
  * Greek planes are transformed one product at a time instead of as one batched
    array (SFFTSubtract.py:628 batches FOMG planes; batching does not change values);
  * Construct_FDIFF evaluates sum_ab a_ijab * Wl^a * Wm^b as a matrix product
    (SFFTConfigure.py:1337-1357 evaluates the same sum per pixel).
"""
import numpy as np

try:  # threaded pocketfft; identical arithmetic to numpy.fft
    import scipy.fft as _sfft

    def _fft2(x, workers=1):
        return _sfft.fft2(x, workers=workers)

    def _ifft2(x, workers=1):
        return _sfft.ifft2(x, workers=workers)
except Exception:  # pragma: no cover
    def _fft2(x, workers=1):
        return np.fft.fft2(x)

    def _ifft2(x, workers=1):
        return np.fft.ifft2(x)


# ----------------------------------------------------------------------------------------------
# SSC: parameter dictionary (SFFTConfigure.py:825-883; Cupy validation :19-28)
# ----------------------------------------------------------------------------------------------
def SSC(NX, NY, KerHW, KerPolyOrder=2, BGPolyOrder=2, ConstPhotRatio=True):
    N0, N1 = int(NX), int(NY)
    w0, w1 = int(KerHW), int(KerHW)
    DK, DB = int(KerPolyOrder), int(BGPolyOrder)
    if DK not in [0, 1, 2, 3]:
        raise Exception('MeLOn ERROR: Input KerPolyOrder should be 0/1/2/3!')
    if DB not in [0, 1, 2, 3]:
        raise Exception('MeLOn ERROR: Input BGPolyOrder should be 0/1/2/3!')
    L0, L1 = 2 * w0 + 1, 2 * w1 + 1
    Fab = L0 * L1
    Fij = int((DK + 1) * (DK + 2) / 2)
    Fpq = int((DB + 1) * (DB + 2) / 2)
    SCALE = np.float64(1 / (N0 * N1))
    SCALE_L = np.float64(1 / SCALE)
    d = dict(N0=N0, N1=N1, w0=w0, w1=w1, DK=DK, DB=DB, ConstPhotRatio=ConstPhotRatio,
             L0=L0, L1=L1, Fab=Fab, Fij=Fij, Fpq=Fpq, SCALE=SCALE, SCALE_L=SCALE_L,
             NEQ=Fij * Fab + Fpq, Fijab=Fij * Fab, NEQ_FSfree=Fij * Fab + Fpq - (Fij - 1),
             FOMG=Fij ** 2, FGAM=Fij * Fpq, FTHE=Fij, FPSI=Fpq * Fij, FPHI=Fpq ** 2, FDEL=Fpq)
    return d


# ----------------------------------------------------------------------------------------------
# index tables (SFFTSubtract.py:514-532)
# ----------------------------------------------------------------------------------------------
def index_tables(p):
    DK, DB, w0, w1, L0, L1 = p['DK'], p['DB'], p['w0'], p['w1'], p['L0'], p['L1']
    Fij, Fab, Fijab, NEQ = p['Fij'], p['Fab'], p['Fijab'], p['NEQ']
    REF_pq = np.array([(pp, q) for pp in range(DB + 1) for q in range(DB + 1 - pp)]).astype(np.int32)
    REF_ij = np.array([(i, j) for i in range(DK + 1) for j in range(DK + 1 - i)]).astype(np.int32)
    REF_ab = np.array([(a - w0, b - w1) for a in range(L0) for b in range(L1)]).astype(np.int32)
    SREF_ijab = np.array([(ij, ab) for ij in range(Fij) for ab in range(Fab)]).astype(np.int32)
    ij00 = np.arange(w0 * L1 + w1, Fijab, Fab).astype(np.int32)
    IDX_nFS = None
    if p['ConstPhotRatio']:
        mask = np.ones(NEQ, dtype=bool)
        mask[ij00[1:]] = False
        IDX_nFS = np.where(mask)[0].astype(np.int32)
    return dict(REF_pq=REF_pq, REF_ij=REF_ij, REF_ab=REF_ab, SREF_ijab=SREF_ijab, ij00=ij00, IDX_nFS=IDX_nFS)


# ----------------------------------------------------------------------------------------------
# SpatialCoor + SpatialPoly (SFFTConfigure.py:889-937)
# ----------------------------------------------------------------------------------------------
def spatial_poly(PixA_I, p, T):
    N0, N1 = p['N0'], p['N1']
    CX = ((np.arange(N0, dtype=np.float64) + 1.0) / N0)[:, None] * np.ones((1, N1))
    CY = np.ones((N0, 1)) * ((np.arange(N1, dtype=np.float64) + 1.0) / N1)[None, :]
    Iij = np.empty((p['Fij'], N0, N1), dtype=np.float64)
    Tpq = np.empty((p['Fpq'], N0, N1), dtype=np.float64)
    for ij, (i, j) in enumerate(T['REF_ij']):
        Iij[ij] = PixA_I * (np.power(CX, i) * np.power(CY, j))
    for pq, (pp, q) in enumerate(T['REF_pq']):
        Tpq[pq] = np.power(CX, pp) * np.power(CY, q)
    return Iij, Tpq


def _mod(v, N):
    # Mod_N(rho): fmod then +N if negative (SFFTConfigure.py:978-1000)
    return np.mod(v, N)


# ----------------------------------------------------------------------------------------------
# Linear system (SFFTSubtract.py:613-730; fill rules SFFTConfigure.py:957-1272)
# ----------------------------------------------------------------------------------------------
def establish_system(PixA_I, PixA_J, p, workers=1):
    """Return (LHMAT[NEQ,NEQ], RHb[NEQ]) exactly as ESSN builds them before stripe removal."""
    T = index_tables(p)
    N0, N1 = p['N0'], p['N1']
    Fij, Fpq, Fab, Fijab, NEQ = p['Fij'], p['Fpq'], p['Fab'], p['Fijab'], p['NEQ']
    SCALE, SCALE_L = p['SCALE'], p['SCALE_L']

    Iij, Tpq = spatial_poly(PixA_I, p, T)
    FJ = SCALE * _fft2(PixA_J.astype(np.complex128), workers)
    FI = np.stack([SCALE * _fft2(Iij[k].astype(np.complex128), workers) for k in range(Fij)])
    FT = np.stack([SCALE * _fft2(Tpq[k].astype(np.complex128), workers) for k in range(Fpq)])
    del Iij, Tpq
    CFJ, CFI, CFT = np.conj(FJ), np.conj(FI), np.conj(FT)

    ab = T['REF_ab']
    a_, b_ = ab[:, 0].astype(np.int64), ab[:, 1].astype(np.int64)
    cen = (a_ == 0) & (b_ == 0)                      # centre element of the delta basis
    MODa, MODb = _mod(a_, N0), _mod(b_, N1)          # (a', b')
    MOD_a, MOD_b = _mod(-a_, N0), _mod(-b_, N1)      # (-a, -b)
    MODda = _mod(a_[:, None] - a_[None, :], N0)      # (a'-a)
    MODdb = _mod(b_[:, None] - b_[None, :], N1)      # (b'-b)

    LHMAT = np.empty((NEQ, NEQ), dtype=np.float64)
    RHb = np.empty(NEQ, dtype=np.float64)

    # OMEGA (SFFTSubtract.py:622-636): PreOMG = SCALE * Re[SCALE * DFT(FI[i8j8] * CFI[ij])]
    for i8j8 in range(Fij):
        for ij in range(Fij):
            Pre = (SCALE * _fft2(FI[i8j8] * CFI[ij], workers)).real
            Pre *= SCALE
            P00 = Pre[0, 0]
            Prow = Pre[MODa, MODb]          # indexed by a8b8
            Pcol = Pre[MOD_a, MOD_b]        # indexed by ab
            blk = -Prow[:, None] - Pcol[None, :] + Pre[MODda, MODdb] + P00
            blk[cen, :] = (Pcol - P00)[None, :]          # row centre, col off-centre
            blk[:, cen] = (Prow - P00)[:, None]          # row off-centre, col centre
            blk[np.ix_(cen, cen)] = P00                  # both centre
            LHMAT[i8j8 * Fab:(i8j8 + 1) * Fab, ij * Fab:(ij + 1) * Fab] = blk

    # GAMMA (SFFTSubtract.py:642-655)
    for i8j8 in range(Fij):
        for pq in range(Fpq):
            Pre = (SCALE * _fft2(FI[i8j8] * CFT[pq], workers)).real
            col = Pre[MODa, MODb] - Pre[0, 0]
            col[cen] = Pre[0, 0]
            LHMAT[i8j8 * Fab:(i8j8 + 1) * Fab, Fijab + pq] = col

    # PSI (SFFTSubtract.py:661-674)
    for p8q8 in range(Fpq):
        for ij in range(Fij):
            Pre = (SCALE * _fft2(FT[p8q8] * CFI[ij], workers)).real
            row = Pre[MOD_a, MOD_b] - Pre[0, 0]
            row[cen] = Pre[0, 0]
            LHMAT[Fijab + p8q8, ij * Fab:(ij + 1) * Fab] = row

    # PHI (SFFTSubtract.py:680-694)
    for p8q8 in range(Fpq):
        for pq in range(Fpq):
            Pre = (SCALE * _fft2(FT[p8q8] * CFT[pq], workers)).real
            Pre *= SCALE_L
            LHMAT[Fijab + p8q8, Fijab + pq] = Pre[0, 0]

    # THETA & DELTA (SFFTSubtract.py:701-729)
    for i8j8 in range(Fij):
        Pre = (SCALE * _fft2(CFJ * FI[i8j8], workers)).real
        v = Pre[MODa, MODb] - Pre[0, 0]
        v[cen] = Pre[0, 0]
        RHb[i8j8 * Fab:(i8j8 + 1) * Fab] = v
    for p8q8 in range(Fpq):
        Pre = (SCALE * _fft2(CFJ * FT[p8q8], workers)).real
        Pre *= SCALE_L
        RHb[Fijab + p8q8] = Pre[0, 0]
    return LHMAT, RHb


def solve_system(LHMAT, RHb, p):
    """Stripe removal, gesv solve, solution extension (SFFTSubtract.py:734-755)."""
    T = index_tables(p)
    if not p['ConstPhotRatio']:
        return np.linalg.solve(LHMAT, RHb).astype(np.float64)
    idx = T['IDX_nFS']
    sol_fs = np.linalg.solve(LHMAT[np.ix_(idx, idx)], RHb[idx]).astype(np.float64)
    Solution = np.zeros(p['NEQ'], dtype=np.float64)
    Solution[idx] = sol_fs
    return Solution


# ----------------------------------------------------------------------------------------------
# Subtraction (SFFTSubtract.py:773-807; Construct_FDIFF SFFTConfigure.py:1316-1359)
# ----------------------------------------------------------------------------------------------
def _twiddle_rows(p):
    N0, N1, w0, w1 = p['N0'], p['N1'], p['w0'], p['w1']
    Wl = np.exp((-2j * np.pi / N0) * np.arange(N0, dtype=np.float64))
    Wm = np.exp((-2j * np.pi / N1) * np.arange(N1, dtype=np.float64))
    Wla = np.stack([Wl ** a for a in range(-w0, w0 + 1)])      # [L0, N0]
    Wmb = np.stack([Wm ** b for b in range(-w1, w1 + 1)])      # [L1, N1]
    return Wla, Wmb


def construct_fdiff(FI, FT, FJ, Solution, p):
    """FDIFF via the per-(ij) kernel transfer function evaluated as Wla^T A_ij Wmb."""
    Fij, Fpq, Fab, Fijab, L0, L1 = p['Fij'], p['Fpq'], p['Fab'], p['Fijab'], p['L0'], p['L1']
    SCALE = p['SCALE']
    a_ijab = Solution[:Fijab].astype(np.complex128)
    b_pq = Solution[Fijab:].astype(np.complex128)
    Wla, Wmb = _twiddle_rows(p)
    cen = p['w0'] * L1 + p['w1']
    PVAL = np.zeros_like(FJ)
    for ij in range(Fij):
        A = a_ijab[ij * Fab:(ij + 1) * Fab].reshape(L0, L1)
        S_off = A.sum() - A.reshape(-1)[cen]
        # sum_ab a_ab * SCALE * (Wl^a Wm^b - 1) for ab != centre, + SCALE * a_centre
        FK = SCALE * ((Wla.T @ A) @ Wmb - S_off)
        PVAL += FI[ij] * FK
    for pq in range(Fpq):
        PVAL += b_pq[pq] * FT[pq]
    return FJ - PVAL


def construct_fdiff_literal(FI, FT, FJ, Solution, p):
    """Transcription of the reference loop order (per ab, then ij); small sizes only."""
    T = index_tables(p)
    Fij, Fpq, Fab, Fijab = p['Fij'], p['Fpq'], p['Fab'], p['Fijab']
    w0, w1, SCALE = p['w0'], p['w1'], p['SCALE']
    N0, N1 = p['N0'], p['N1']
    a_ijab = Solution[:Fijab].astype(np.complex128)
    b_pq = Solution[Fijab:].astype(np.complex128)
    X = np.arange(N0, dtype=np.float64)[:, None] * np.ones((1, N1))
    Y = np.ones((N0, 1)) * np.arange(N1, dtype=np.float64)[None, :]
    Wl = np.exp((-2j * np.pi / N0) * X)
    Wm = np.exp((-2j * np.pi / N1) * Y)
    PVAL = np.zeros((N0, N1), dtype=np.complex128)
    for ab in range(Fab):
        a, b = T['REF_ab'][ab]
        if a == 0 and b == 0:
            FKab = (SCALE + 0j) * np.ones((N0, N1), dtype=np.complex128)
        else:
            FKab = (SCALE + 0j) * ((Wl ** int(a)) * (Wm ** int(b)) - 1.0)
        for ij in range(Fij):
            PVAL += (a_ijab[ij * Fab + ab] * FI[ij]) * FKab
    for pq in range(Fpq):
        PVAL += b_pq[pq] * FT[pq]
    return FJ - PVAL


def subtract(PixA_I, PixA_J, Solution, p, workers=1, literal=False):
    T = index_tables(p)
    SCALE, SCALE_L, Fij, Fpq = p['SCALE'], p['SCALE_L'], p['Fij'], p['Fpq']
    Iij, Tpq = spatial_poly(PixA_I, p, T)
    FJ = SCALE * _fft2(PixA_J.astype(np.complex128), workers)
    FI = np.stack([SCALE * _fft2(Iij[k].astype(np.complex128), workers) for k in range(Fij)])
    FT = np.stack([SCALE * _fft2(Tpq[k].astype(np.complex128), workers) for k in range(Fpq)])
    fn = construct_fdiff_literal if literal else construct_fdiff
    FDIFF = fn(FI, FT, FJ, Solution.astype(np.float64), p)
    DIFF = (SCALE_L * _ifft2(FDIFF, workers)).real
    return np.ascontiguousarray(DIFF)


# ----------------------------------------------------------------------------------------------
# ESS / GSS / packet level
# ----------------------------------------------------------------------------------------------
def ESS(PixA_I, PixA_J, p, SFFTSolution=None, Subtract=False, workers=1, literal=False):
    N0, N1 = p['N0'], p['N1']
    if PixA_I.shape != (N0, N1) or PixA_J.shape != (N0, N1):
        raise Exception('MeLOn ERROR: INCONSISTENT shape of input images I & J, [%d, %d] required!' % (N0, N1))
    PixA_I = np.ascontiguousarray(PixA_I, np.float64)
    PixA_J = np.ascontiguousarray(PixA_J, np.float64)
    if SFFTSolution is None:
        LHMAT, RHb = establish_system(PixA_I, PixA_J, p, workers)
        Solution = solve_system(LHMAT, RHb, p)
    else:
        Solution = np.asarray(SFFTSolution, dtype=np.float64)
    PixA_DIFF = None
    if Subtract:
        PixA_DIFF = subtract(PixA_I, PixA_J, Solution, p, workers, literal)
    return Solution, PixA_DIFF


def GSS(PixA_I, PixA_J, PixA_mI, PixA_mJ, p, ContamMask_I=None, workers=1):
    tmplst = [PixA_I.shape, PixA_J.shape, PixA_mI.shape, PixA_mI.shape]   # sic (SFFTSubtract.py:892)
    if len(set(tmplst)) > 1:
        raise Exception('MeLOn ERROR: Input images should have same size!')
    Solution = ESS(PixA_mI, PixA_mJ, p, None, False, workers)[0]
    PixA_DIFF = ESS(PixA_I, PixA_J, p, Solution, True, workers)[1]
    ContamMask_CI = None
    if ContamMask_I is not None:                                           # SFFTSubtract.py:907-921
        tSolution = Solution.copy()
        tSolution[-p['Fpq']:] = 0.0
        _tmpD = ESS(ContamMask_I.astype(np.float64), np.zeros(PixA_J.shape), p, tSolution, True, workers)[1]
        ContamMask_CI = _tmpD < -0.001
    return Solution, PixA_DIFF, ContamMask_CI


def CP_arrays(PixA_REF, PixA_SCI, PixA_mREF, PixA_mSCI, ForceConv, GKerHW,
              KerPolyOrder=2, BGPolyOrder=2, ConstPhotRatio=True, workers=1):
    """Array-level body of Customized_Packet.CP (CustomizedPacket.py:114-188)."""
    NaNmask_U = None
    nr, ns = np.isnan(PixA_REF), np.isnan(PixA_SCI)
    if nr.any() or ns.any():
        NaNmask_U = np.logical_or(nr, ns)
    assert np.sum(np.isnan(PixA_mREF)) == 0
    assert np.sum(np.isnan(PixA_mSCI)) == 0
    assert ForceConv in ['REF', 'SCI']
    p = SSC(PixA_REF.shape[0], PixA_REF.shape[1], GKerHW, KerPolyOrder, BGPolyOrder, ConstPhotRatio)
    if ForceConv == 'REF':
        mI, mJ, I, J = PixA_mREF, PixA_mSCI, PixA_REF, PixA_SCI
    else:
        mI, mJ, I, J = PixA_mSCI, PixA_mREF, PixA_SCI, PixA_REF
    if NaNmask_U is not None:
        I, J = I.copy(), J.copy()
        I[NaNmask_U] = mI[NaNmask_U]
        J[NaNmask_U] = mJ[NaNmask_U]
    Solution, DIFF, _ = GSS(I, J, mI, mJ, p, None, workers)
    if NaNmask_U is not None:
        DIFF[NaNmask_U] = np.nan
    if ForceConv == 'SCI':
        DIFF = -DIFF
    return Solution, DIFF
