"""Pins for oracle/bspline_sv_oracle.py (SEPARATE-VARYING scaling + kernel regularisation of sfft/BSplineSFFT.py).

The reference has no CPU implementation of these modes; its one artefact that went through them, the NIRCam example's SNR map, pins
the oracle end to end (tests/test_nircam_chain.py); these tests pin the modes that example does not use.
These tests pin the restatement from three independent sides: brute-force normal equations, the reduction to the
ENTANGLED system that IS pinned by goldens, and structural properties of the regularisation matrix."""
import numpy as np
import pytest

from oracle import bspline_oracle as bo
from oracle import bspline_sv_oracle as sv
from sfft_amd.utils.synthetic import make_pair


def _case(N0=24, N1=20, w=1, ker=('Polynomial', 2, (), ()), sca=('Polynomial', 1, (), ()), bkg=('Polynomial', 1, (), ()), seed=3):
    pair = make_pair(N0, N1, seed=seed, density=60.0)
    I, J = pair['REF'], pair['SCI']
    basis = bo.make_basis(N0, N1, ker[0], ker[1], ker[2], ker[3], bkg[0], bkg[1], bkg[2], bkg[3])
    Fij = len(basis['ker_pairs'])
    scab = sv.make_scaling_basis(N0, N1, Fij, sca[0], sca[1], sca[2], sca[3])
    p = sv.SSC(N0, N1, w, basis, scab, 'SEPARATE-VARYING')
    return I, J, basis, scab, p


CASES = [
    dict(),                                                                          # poly kernel 2, poly scaling 1
    dict(ker=('B-Spline', 2, (12.5,), ()), sca=('B-Spline', 1, (), ()), w=2, N0=28, N1=24),
    dict(ker=('B-Spline', 1, (10.5,), (9.5,)), sca=('Polynomial', 2, (), ()), bkg=('B-Spline', 1, (), ())),
]


@pytest.mark.parametrize("kw", CASES)
def test_system_is_the_normal_equations_of_the_model(kw):
    I, J, basis, scab, p = _case(**kw)
    LHMAT, RHb = sv.establish_system(I, J, p, basis, scab)
    A = sv.design_matrix(I, p, basis, scab)
    ref_L = p['SCALE'] * (A.T @ A)
    ref_b = p['SCALE'] * (A.T @ J.reshape(-1))
    assert np.abs(LHMAT - ref_L).max() <= 1e-10 * np.abs(ref_L).max()
    assert np.abs(RHb - ref_b).max() <= 1e-10 * np.abs(ref_b).max()
    # place-holder scaling terms give exactly empty rows / columns
    ij00 = np.arange(p['w0'] * p['L1'] + p['w1'], p['Fijab'], p['Fab'])
    dead = ij00[p['ScaFij']:]
    assert np.all(LHMAT[dead, :] == 0) and np.all(LHMAT[:, dead] == 0) and np.all(RHb[dead] == 0)


def test_reduces_to_entangled_when_scaling_basis_is_kernel_basis():
    N0, N1, w = 32, 24, 2
    pair = make_pair(N0, N1, seed=5, density=60.0)
    I, J = pair['REF'], pair['SCI']
    basis = bo.make_basis(N0, N1, 'B-Spline', 2, (16.5,), (), 'Polynomial', 1, (), ())
    Fij = len(basis['ker_pairs'])
    scab = sv.make_scaling_basis(N0, N1, Fij, 'B-Spline', 2, (16.5,), ())
    assert scab['ScaFij'] == Fij
    p = sv.SSC(N0, N1, w, basis, scab, 'SEPARATE-VARYING')
    assert p['NEQt'] == p['NEQ']
    L1, b1 = sv.establish_system(I, J, p, basis, scab)
    pe = bo.SSC(N0, N1, w, basis, ConstPhotRatio=False)
    L0, b0 = bo.establish_system(I, J, pe, basis)
    assert np.abs(L1 - L0).max() <= 1e-13 * np.abs(L0).max()
    assert np.abs(b1 - b0).max() <= 1e-13 * np.abs(b0).max()
    sol1 = sv.solve_system(L1, b1, p)
    d1 = sv.subtract(I, J, sol1, p, basis, scab)
    d0 = bo.subtract(I, J, sol1, pe, basis)
    assert np.abs(d1 - d0).max() <= 1e-9 * np.abs(J).max()


@pytest.mark.parametrize("kw", CASES)
def test_solution_recovers_a_varying_scaling_and_diff_is_the_residual(kw):
    I, J, basis, scab, p = _case(**kw)
    # science image built from the model itself: kernel = delta with scaling s(x, y), plus background
    rng = np.random.default_rng(11)
    truth = np.zeros(p['NEQ'])
    ij00 = np.arange(p['w0'] * p['L1'] + p['w1'], p['Fijab'], p['Fab'])
    truth[ij00[:p['ScaFij']]] = rng.uniform(0.5, 1.5, p['ScaFij']) * p['SCALE_L']
    truth[p['Fijab']:] = rng.uniform(-2, 2, p['Fpq'])
    A = sv.design_matrix(I, p, basis, scab)
    Jm = (A @ truth).reshape(I.shape)
    Solution, DIFF = sv.ESS(I, Jm, p, basis, scab, None, True)
    assert np.abs(DIFF).max() <= 1e-7 * np.abs(Jm).max()
    keep = np.setdiff1d(np.arange(p['NEQ']), ij00[p['ScaFij']:])
    resid = (A[:, keep] @ Solution[keep]).reshape(I.shape) - Jm
    assert np.abs(resid).max() <= 1e-7 * np.abs(Jm).max()
    assert np.all(Solution[ij00[p['ScaFij']:]] == 0)


def test_subtract_equals_real_space_model():
    I, J, basis, scab, p = _case()
    rng = np.random.default_rng(2)
    sol = rng.normal(size=p['NEQ'])
    ij00 = np.arange(p['w0'] * p['L1'] + p['w1'], p['Fijab'], p['Fab'])
    sol[ij00[p['ScaFij']:]] = 0.0
    sol[:p['Fijab']] *= p['SCALE_L']
    A = sv.design_matrix(I, p, basis, scab)
    DIFF = sv.subtract(I, J, sol, p, basis, scab)
    ref = J - (A @ sol).reshape(I.shape)
    assert np.abs(DIFF - ref).max() <= 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("mode", ['ENTANGLED', 'SEPARATE-VARYING'])
def test_regularisation_matrix_properties(mode):
    N0, N1, w = 24, 20, 2
    basis = bo.make_basis(N0, N1, 'Polynomial', 2, (), (), 'Polynomial', 1, (), ())
    Fij = len(basis['ker_pairs'])
    scab = sv.make_scaling_basis(N0, N1, Fij, 'Polynomial', 1) if mode == 'SEPARATE-VARYING' else None
    p = sv.SSC(N0, N1, w, basis, scab, mode)
    XY = np.array([[x, y] for x in (3.0, 12.0, 21.0) for y in (4.0, 10.0, 17.0)])
    W = np.linspace(1.0, 2.0, XY.shape[0])
    iREG = sv.laplacian_ireg(w, w, IGNORE_LAPLACIAN_KERCENT=False)
    assert np.array_equal(iREG, iREG.T)
    kerspec = dict(KerSpType='Polynomial', DK=2, KerIntKnotX=[], KerIntKnotY=[])
    SST, CSST, DSST = sv.spatial_gram(p, kerspec, scab, XY, W)
    REG = sv.regularization_matrix(p, iREG, SST, CSST, DSST)
    assert np.abs(REG - REG.T).max() <= 1e-15 * np.abs(REG).max()
    ev = np.linalg.eigvalsh(REG)
    assert ev.min() >= -1e-12 * ev.max()
    # the penalty is  sum_k W_k |Laplacian of the kernel stamp at (x_k, y_k)|^2 (both columns of the symmetrised form):
    # check against a direct evaluation for a random solution
    rng = np.random.default_rng(4)
    sol = rng.normal(size=p['NEQ'])
    L = 2 * w + 1
    c0 = w * L + w
    LAP = np.zeros((L * L, L * L))
    for r in range(L * L):
        for c in range(L * L):
            dr, dc = abs(r // L - c // L), abs(r % L - c % L)
            if r == c:
                LAP[r, c] = sum(1 for (u, v) in ((-1, 0), (1, 0), (0, -1), (0, 1)) if 0 <= r // L + u < L and 0 <= r % L + v < L)
            elif dr + dc == 1:
                LAP[r, c] = -1
    CX, CY = XY[:, 0] / N0, XY[:, 1] / N1
    SP = np.array([CX ** i * CY ** j for i in range(3) for j in range(3 - i)])
    if mode == 'SEPARATE-VARYING':
        ScaSP = np.array([CX ** i * CY ** j for i in range(2) for j in range(2 - i)])
        ScaSP = np.concatenate((ScaSP, np.zeros((Fij - ScaSP.shape[0], XY.shape[0]))))
    else:
        ScaSP = SP
    pen = 0.0
    a = sol[:p['Fijab']].reshape(Fij, L * L)
    for k in range(XY.shape[0]):
        # standard kernel stamp at point k: off-centre pixels sum_ij a_ijab B_ij, centre = scaling - sum(off-centre)
        off = (a * SP[:, k][:, None]).sum(axis=0)
        off[c0] = 0.0
        stamp = off.copy()
        stamp[c0] = (a[:, c0] * ScaSP[:, k]).sum() - off.sum()
        pen += (W[k] / W.sum()) * np.sum((LAP @ stamp) ** 2)
    quad = sol @ REG @ sol / p['SCALE'] ** 2
    assert abs(quad - 2.0 * pen) <= 1e-10 * abs(quad)


def test_lambda_zero_is_no_regularisation_and_large_lambda_smooths():
    I, J, basis, scab, p = _case(w=2)
    iREG = sv.laplacian_ireg(2, 2, True)
    kerspec = dict(KerSpType='Polynomial', DK=2, KerIntKnotX=[], KerIntKnotY=[])
    XY = np.array([[6.0, 5.0], [18.0, 15.0], [12.0, 10.0]])
    SST, CSST, DSST = sv.spatial_gram(p, kerspec, scab, XY, None)
    REG = sv.regularization_matrix(p, iREG, SST, CSST, DSST)
    s0 = sv.ESS(I, J, p, basis, scab)[0]
    s1 = sv.ESS(I, J, p, basis, scab, REGMAT=REG, LAMBDA_REGULARIZE=0.0)[0]
    assert np.array_equal(s0, s1)
    L0, b0 = sv.establish_system(I, J, p, basis, scab)
    lam = 1e3 * np.abs(L0).max() / np.abs(REG).max()
    s2 = sv.ESS(I, J, p, basis, scab, REGMAT=REG, LAMBDA_REGULARIZE=lam)[0]
    assert s2 @ REG @ s2 < 1e-3 * (s0 @ REG @ s0)


# ---------------------------------------------------------------------------------------------------
# Two transcriptions must agree: oracle/bspline_sv_literal.py follows the reference's CUDA kernels and host loops line by
# line (per-element loops, the same index tables and cIdx loops); oracle/bspline_sv_oracle.py is the vectorised restatement
# the GPU tests use.  Agreement to 1e-13 guards the vectorisation; the reference-made pin is tests/test_nircam_chain.py.
# ---------------------------------------------------------------------------------------------------
from oracle import bspline_sv_literal as lit

LIT_CASES = [
    dict(N0=16, N1=12, w=1),                                                                                     # poly kernel 2, poly scaling 1
    dict(N0=18, N1=14, w=1, ker=('B-Spline', 1, (9.5,), ()), sca=('Polynomial', 1, (), ()), bkg=('Polynomial', 0, (), ())),
    dict(N0=14, N1=16, w=2, ker=('B-Spline', 2, (), ()), sca=('B-Spline', 1, (), ()), bkg=('B-Spline', 1, (), ())),
]


def _lit_args(I, J, basis, scab, p):
    return (I, J, p['N0'], p['N1'], p['w0'], p['w1'], basis['kbx'], basis['kby'], [tuple(x) for x in basis['ker_pairs']],
            scab['sbx'], scab['sby'], [tuple(x) for x in scab['sca_pairs']], basis['tbx'], basis['tby'],
            [tuple(x) for x in basis['bkg_pairs']])


@pytest.mark.parametrize("kw", LIT_CASES)
def test_literal_transcription_agrees_with_the_vectorised_oracle(kw):
    I, J, basis, scab, p = _case(**kw)
    L_v, b_v = sv.establish_system(I, J, p, basis, scab)
    L_l, b_l = lit.establish_system(*_lit_args(I, J, basis, scab, p))
    nk = p['Fijab']
    for blk_v, blk_l in ((L_v[:nk, :nk], L_l[:nk, :nk]), (L_v[:nk, nk:], L_l[:nk, nk:]), (L_v[nk:, :nk], L_l[nk:, :nk]),
                         (L_v[nk:, nk:], L_l[nk:, nk:]), (b_v[:nk], b_l[:nk]), (b_v[nk:], b_l[nk:])):
        assert np.abs(blk_v - blk_l).max() <= 1e-13 * np.abs(blk_l).max()
    # regularisation: Laplacian -> iREGMAT -> REGMAT, unweighted and weighted
    XY = np.array([[3.0, 2.5], [8.0, 9.0], [12.5, 4.0], [5.5, 10.5]])
    kerspec = dict(KerSpType=basis['KerSpType'], DK=kw.get('ker', ('Polynomial', 2))[1],
                   KerIntKnotX=list(kw.get('ker', ('', 0, (), ()))[2]), KerIntKnotY=list(kw.get('ker', ('', 0, (), ()))[3]))
    for W in (None, np.array([1.0, 2.0, 0.5, 3.0])):
        SST, CSST, DSST = sv.spatial_gram(p, kerspec, scab, XY, W)
        REG_v = sv.regularization_matrix(p, sv.laplacian_ireg(p['w0'], p['w1'], True), SST, CSST, DSST)
        REG_l, iREG_l = lit.regularization_matrix(p['N0'], p['N1'], p['w0'], p['w1'], p['Fij'], p['Fpq'], SST, CSST, DSST, True)
        assert np.array_equal(iREG_l, sv.laplacian_ireg(p['w0'], p['w1'], True))
        assert np.abs(REG_v - REG_l).max() <= 1e-13 * np.abs(REG_l).max()
    # tweak + solve + restore, then the difference image, both from the literal system
    lam = 10.0 / p['SCALE'] ** 2 * 1e-6
    sol_v = sv.solve_system(L_l + lam * REG_l, b_l, p)
    sol_l = lit.tweak_solve_restore(L_l + lam * REG_l, b_l, p['Fij'], p['Fpq'], p['w0'], p['w1'], 'SEPARATE-VARYING',
                                    basis['KerSpType'], ScaFij=p['ScaFij'])
    assert np.abs(sol_v - sol_l).max() <= 1e-9 * np.abs(sol_l).max()
    ij00 = np.arange(p['w0'] * p['L1'] + p['w1'], p['Fijab'], p['Fab'])
    assert np.all(sol_l[ij00[p['ScaFij']:]] == 0.0)
    D_v = sv.subtract(I, J, sol_l, p, basis, scab)
    D_l = lit.construct_diff(*((I, J, sol_l) + _lit_args(I, J, basis, scab, p)[2:]))
    assert np.sqrt(np.mean((D_v - D_l) ** 2)) <= 1e-12 * np.sqrt(np.mean(J ** 2))


def test_literal_spatial_grams_and_constant_scaling_tweak_agree():
    """spatial Gram matrices (:3583-3635) and the B-spline TweakLS of SEPARATE-CONSTANT scaling (the sum rule, :2201-2272)."""
    rng = np.random.default_rng(1)
    SP, ScaSP = rng.normal(size=(6, 5)), rng.normal(size=(3, 5))
    for W in (None, np.array([1.0, 2.0, 3.0, 4.0, 5.0])):
        SST, CSST, DSST = lit.spatial_grams(SP, ScaSP, 6, W)
        Wm = np.eye(5) / 5 if W is None else np.diag(W) / W.sum()
        Sp = np.concatenate((ScaSP, np.zeros((3, 5))), axis=0)
        assert np.allclose(SST, SP @ Wm @ SP.T, rtol=1e-14, atol=0) and np.allclose(CSST, SP @ Wm @ Sp.T, rtol=1e-14, atol=1e-300)
        assert np.allclose(DSST, Sp @ Wm @ Sp.T, rtol=1e-14, atol=1e-300)
    N0, N1, w = 20, 16, 1
    pair = make_pair(N0, N1, seed=9, density=60.0)
    basis = bo.make_basis(N0, N1, 'B-Spline', 1, (10.5,), (), 'Polynomial', 1, (), ())
    pc = bo.SSC(N0, N1, w, basis, ConstPhotRatio=True)
    LH, rhs = bo.establish_system(pair['REF'], pair['SCI'], pc, basis)
    sol_o = bo.solve_system(LH, rhs, pc)                                   # pinned by reference-made goldens
    sol_l = lit.tweak_solve_restore(LH, rhs, pc['Fij'], pc['Fpq'], w, w, 'SEPARATE-CONSTANT', 'B-Spline')
    assert np.abs(sol_o - sol_l).max() <= 1e-9 * np.abs(sol_o).max()
    ij00 = np.arange(w * 3 + w, pc['Fijab'], pc['Fab'])
    assert np.all(sol_l[ij00] == sol_l[ij00[0]])


def test_gridconvolve_oracle_uses_convolve2d_as_defined():
    """oracle/gridconv_oracle.py loops over segments like the reference (BSplineSFFT.py:4985-5003) and calls scipy's convolve2d; here
    that call is checked against the definition of a 'same', zero-filled convolution written out as four loops."""
    from scipy.signal import convolve2d
    from oracle import gridconv_oracle as gc
    rng = np.random.default_rng(4)
    img, ker = rng.normal(size=(13, 11)), rng.normal(size=(5, 3))
    assert np.abs(convolve2d(img, ker, mode='same', boundary='fill', fillvalue=0.0) - lit.convolve2d_same_fill0(img, ker)).max() <= 1e-13
    # ... and the segment loop against a per-pixel evaluation with the pixel's own segment kernel
    N0, N1, TiHW = 17, 14, 2
    AllocatedL, _ = gc.tile_labels(N0, N1, TiHW)
    Nseg = AllocatedL.max() + 1
    KerStack = rng.uniform(0.5, 1.5, size=(Nseg, 3, 3))
    img = rng.normal(size=(N0, N1))
    out = gc.gsvc(img, AllocatedL, KerStack, normalize_kernel=True)
    ref = np.zeros((N0, N1))
    for x in range(N0):
        for y in range(N1):
            K = KerStack[AllocatedL[x, y]] / KerStack[AllocatedL[x, y]].sum()
            acc = 0.0
            for u in range(3):
                for v in range(3):
                    xx, yy = x + 1 - u, y + 1 - v
                    if 0 <= xx < N0 and 0 <= yy < N1:
                        acc += K[u, v] * img[xx, yy]
            ref[x, y] = acc
    assert np.abs(out - ref).max() <= 1e-13
