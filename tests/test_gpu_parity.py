"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI against
  (a) golden vectors produced by the reference's own Numpy backend (tests/golden/*.npz),
  (b) the CPU oracle on seeded inputs at sizes it finishes in seconds,
  (c) size-independent properties at the BASELINE size (4096 x 4096, KerHW 8, orders 2/2).

Tolerances (fp64; SURVEY.md 8c, restated in DESIGN.md):
  forward spectra            max |err| <= 1e-12 * max|spectrum|
  LHMAT / RHb                element-wise <= 1e-11 * max|block|
  apply-only DIFF            pixel RMS error <= 1e-10 * RMS(J)
  end-to-end DIFF            pixel RMS error <= 1e-6  * RMS(DIFF_ref)
"""
import os

import numpy as np
import pytest

from _golden import golden_names, load_golden, packet_roles, rms, rel_rms_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from sfft_amd import _lib
    _lib.lib()          # fail loudly if the HIP library is missing
    return torch.device("cuda", 0)


def _to(dev, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)


def _plan(meta, dev):
    from sfft_amd.plan import get_plan
    return get_plan(meta["N0"], meta["N1"], meta["KerHW"], meta["DK"], meta["DB"], bool(meta["CPR"]), dev.index)


NAMES = golden_names()


# ------------------------------------------------------------------------------------------------
# (0) building blocks
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(8, 8), (64, 64), (128, 32), (48, 40), (45, 35), (100, 96), (512, 256), (300, 500),
                                   (1024, 2048), (4096, 64), (64, 4096), (4096, 4096), (4095, 4096), (8192, 16),
                                   (6144, 40), (40, 6144), (6144, 6144), (9232, 64), (64, 9216), (4100, 5000), (4097, 24), (16, 12261), (4513, 32), (40, 4603),     # big axes (12261 = 3 * 61 * 67); primes up to 4608 through Bluestein on 9216 points
                                   (12, 24), (9, 18), (27, 24), (81, 162), (96, 1536), (1536, 96), (8748, 16), (16, 8748),
                                   (9216, 24), (2304, 3072),    # 2^a*3^b axes (mixed-radix on-chip transform)
                                   (4621, 24), (24, 4621), (10006, 16), (16, 10007), (10007, 40), (5003, 4621)])    # prime factors above the on-chip
                                   # Bluestein limit (4621 and 10007 are prime, 10006 = 2 x 5003): Bluestein through a 16384- / 32768-point four-step transform
@pytest.mark.parametrize("ij", [(0, 0), (2, 1)])
def test_forward_spectrum_matches_numpy_fft2(dev, shape, ij):
    from sfft_amd.plan import get_plan
    N0, N1 = shape
    rng = np.random.default_rng(N0 * 7 + N1)
    img = rng.normal(size=shape) * 50 + 10
    plan = get_plan(N0, N1, 1, 2, 0, True, dev.index)
    F = plan.forward_spectrum(_to(dev, img), ij[0], ij[1]).cpu().numpy()
    cx = ((np.arange(N0) + 1.0) / N0)[:, None]
    cy = ((np.arange(N1) + 1.0) / N1)[None, :]
    ref = (np.fft.fft2(img * (cx ** ij[0] * cy ** ij[1])) / (N0 * N1))[:, :N1 // 2 + 1]
    assert np.max(np.abs(F - ref)) <= 1e-12 * np.max(np.abs(ref))


# ------------------------------------------------------------------------------------------------
# (a) golden vectors from the reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", NAMES)
def test_linear_system_matches_reference(dev, name):
    g = load_golden(name)
    plan = _plan(g["meta"], dev)
    I, J, mI, mJ, _ = packet_roles(g)
    plan.solve(_to(dev, mI), _to(dev, mJ))
    LH, rhs = plan.get_system()
    LH, rhs = LH.cpu().numpy(), rhs.cpu().numpy()
    assert np.max(np.abs(LH - g["LHMAT"])) <= 1e-11 * np.max(np.abs(g["LHMAT"]))
    assert np.max(np.abs(rhs - g["RHb"])) <= 1e-11 * np.max(np.abs(g["RHb"]))
    # ... and the system the factorisation itself received (same fill_system launch, caller's buffer): the reference's
    # LHMAT_FSfree / RHb_FSfree = LHMAT, RHb with the rows / columns ij00[1:] removed (Remove_LSFStripes, SFFTSubtract.py:386-394)
    m = g["meta"]
    A, b, idx = (t.cpu().numpy() for t in plan.get_solver_system())
    L = 2 * m["KerHW"] + 1
    Fij = (m["DK"] + 1) * (m["DK"] + 2) // 2
    ij00 = np.arange(m["KerHW"] * L + m["KerHW"], Fij * L * L, L * L)
    keep = np.setdiff1d(np.arange(g["LHMAT"].shape[0]), ij00[1:]) if bool(m["CPR"]) else np.arange(g["LHMAT"].shape[0])
    assert np.array_equal(idx, keep)
    assert np.max(np.abs(A - g["LHMAT"][np.ix_(keep, keep)])) <= 1e-11 * np.max(np.abs(g["LHMAT"]))
    assert np.max(np.abs(b - g["RHb"][keep])) <= 1e-11 * np.max(np.abs(g["RHb"]))


@pytest.mark.parametrize("name", NAMES)
def test_apply_given_reference_solution(dev, name):
    g = load_golden(name)
    plan = _plan(g["meta"], dev)
    I, J, mI, mJ, nm = packet_roles(g)
    DIFF = plan.apply(_to(dev, I), _to(dev, J), _to(dev, g["Solution"])).cpu().numpy()
    if nm is not None:
        DIFF[nm] = np.nan
    if g["meta"]["ForceConv"] == "SCI":
        DIFF = -DIFF
    assert rms(DIFF - g["DIFF"]) <= 1e-10 * rms(J)


@pytest.mark.parametrize("name", NAMES)
def test_pccp_end_to_end_matches_reference(dev, name):
    from sfft_amd import PureCupy_Customized_Packet
    g = load_golden(name)
    m = g["meta"]
    sol, diff = PureCupy_Customized_Packet.PCCP(_to(dev, g["REF"]), _to(dev, g["SCI"]), _to(dev, g["mREF"]),
                                                _to(dev, g["mSCI"]), m["ForceConv"], m["KerHW"], KerPolyOrder=m["DK"],
                                                BGPolyOrder=m["DB"], ConstPhotRatio=bool(m["CPR"]),
                                                CUDA_DEVICE_4SUBTRACT=str(dev.index), VERBOSE_LEVEL=0)
    DIFF = diff.cpu().numpy()
    assert np.array_equal(np.isnan(DIFF), np.isnan(g["DIFF"]))
    assert rel_rms_err(DIFF, g["DIFF"]) <= 1e-6
    sol = sol.cpu().numpy()
    assert sol.shape == g["Solution"].shape
    if bool(m["CPR"]) and m["DK"] > 0:      # forbidden stripes stay exactly zero (Extend_Solution)
        Fab = (2 * m["KerHW"] + 1) ** 2
        cen = m["KerHW"] * (2 * m["KerHW"] + 1) + m["KerHW"]
        Fij = (m["DK"] + 1) * (m["DK"] + 2) // 2
        assert all(sol[ij * Fab + cen] == 0.0 for ij in range(1, Fij))


@pytest.mark.parametrize("name", ["c64x64_w2_k2b2_cpr", "c45x35_w2_k1b2_free", "c96x80_w3_k2b2_cpr_nan"])
def test_lu_fallback_matches_reference(dev, name):
    g = load_golden(name)
    plan = _plan(g["meta"], dev)
    I, J, mI, mJ, nm = packet_roles(g)
    plan.set_force_lu(True)
    try:
        sol, diff = plan.subtract(_to(dev, I), _to(dev, J), _to(dev, mI), _to(dev, mJ))
        assert plan.query("LAST_SOLVER") == 2
    finally:
        plan.set_force_lu(False)
    DIFF = diff.cpu().numpy()
    if nm is not None:
        DIFF[nm] = np.nan
    if g["meta"]["ForceConv"] == "SCI":
        DIFF = -DIFF
    assert rel_rms_err(DIFF, g["DIFF"]) <= 1e-6


def test_customized_packet_fits_files(dev, tmp_path):
    """File-based operator (Customized_Packet.CP) on FITS written with the minimal FITS module."""
    from sfft_amd import Customized_Packet
    from sfft_amd.utils import minifits
    g = load_golden("c64x32_w3_k2b0_cpr")
    m = g["meta"]
    paths = {}
    for k in ("REF", "SCI", "mREF", "mSCI"):
        paths[k] = str(tmp_path / (k + ".fits"))
        minifits.writeto(paths[k], np.ascontiguousarray(g[k].T))     # FITS axes are transposed (CustomizedPacket.py:93)
    fdiff, fsol = str(tmp_path / "diff.fits"), str(tmp_path / "sol.fits")
    sol, diff = Customized_Packet.CP(paths["REF"], paths["SCI"], paths["mREF"], paths["mSCI"], m["ForceConv"], m["KerHW"],
                                     FITS_DIFF=fdiff, FITS_Solution=fsol, KerPolyOrder=m["DK"], BGPolyOrder=m["DB"],
                                     ConstPhotRatio=bool(m["CPR"]), BACKEND_4SUBTRACT="Cupy",
                                     CUDA_DEVICE_4SUBTRACT=str(dev.index), VERBOSE_LEVEL=0)
    assert isinstance(sol, np.ndarray) and isinstance(diff, np.ndarray)
    assert rel_rms_err(diff, g["DIFF"]) <= 1e-6
    d2, cards = minifits.getdata(fdiff)
    h = minifits.header_dict(cards)
    assert h["KERHW"] == m["KerHW"] and h["CONVD"] == m["ForceConv"] and h["KERORDER"] == m["DK"]
    assert np.allclose(d2.T, diff, rtol=0, atol=0)
    s2, c2 = minifits.getdata(fsol)
    assert s2.shape == (1, sol.size) and np.array_equal(s2[0], sol)
    assert minifits.header_dict(c2)["FIJAB"] == (2 * m["KerHW"] + 1) ** 2 * 6


# ------------------------------------------------------------------------------------------------
# (b) oracle on seeded inputs
# ------------------------------------------------------------------------------------------------
ORACLE_CASES = [
    # N0, N1, w, DK, DB, CPR, ForceConv, mask
    (256, 256, 4, 2, 2, True, "REF", True),
    (384, 200, 3, 1, 3, True, "SCI", False),
    (200, 333, 2, 3, 0, False, "REF", True),
    (512, 512, 8, 2, 2, True, "REF", True),     # BASELINE kernel geometry at a size the oracle finishes in seconds
    (128, 128, 0, 2, 1, True, "REF", False),    # degenerate 1x1 kernel
    (64, 48, 9, 0, 0, True, "REF", False),      # 4w+1 > N1/2: lags wrap around the image
    (6144, 72, 2, 1, 1, True, "REF", False),    # axis 0 needs the four-step transform (6144 = 2048 x 3)
    (80, 9232, 2, 1, 0, True, "SCI", True),     # axis 1 needs it (9232 = 16 x 577, Bluestein inside)
    (160, 144, 2, 3, 2, False, "REF", False),   # NEQ = 256 = 4 x 64: the fused Cholesky steps end exactly at the matrix edge
    (160, 144, 2, 1, 0, True, "REF", False),    # NEQ_FSfree = 74: one fused step would not fit (n < 128), two-kernel path only
    # KerHW > 8: Omega lag half-width 2 w > 16 on the matrix cores, 16 lags per launch (lag0 = 0, 16)
    (208, 160, 10, 1, 1, True, "REF", True),    # h = 20, N0 a multiple of 16: decimated launches, the second one needs 4 of its 16 lags
    (192, 176, 12, 1, 1, True, "SCI", True),    # h = 24 (config 5's KerHW): decimated, second launch at half the matrix instructions
    (130, 140, 16, 1, 0, False, "REF", True),   # h = 32, N0 not a multiple of 16: undecimated launches with the row mask, both full
    (256, 96, 14, 0, 1, True, "REF", False),    # h = 28, a single kernel plane: one diagonal pass, no groups of three
    # sides with a prime factor above the on-chip Bluestein limit (round 5): Bluestein through the four-step transform, forward AND inverse,
    # column axis / row axis, the whole subtraction against the oracle
    (4621, 40, 2, 1, 1, True, "REF", False),    # 4621 is prime: 16384-point four-step transforms on the column axis
    (48, 10006, 2, 2, 1, False, "SCI", True),   # 10006 = 2 x 5003 on the row axis (packed row pairs): 32768-point transforms
]


@pytest.mark.parametrize("case", ORACLE_CASES)
def test_gss_matches_oracle(dev, case):
    from oracle import sfft_oracle as O
    from sfft_amd import PureCupy_Customized_Packet
    from sfft_amd.utils.synthetic import make_pair
    N0, N1, w, DK, DB, CPR, FC, mask = case
    pair = make_pair(N0, N1, seed=N0 + 3 * N1 + w, mask=mask, sky=0.0 if mask else 100.0, bkg_scale=0.05 if mask else 1.0)
    sol_o, diff_o = O.CP_arrays(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], FC, w, DK, DB, CPR, workers=8)
    sol, diff = PureCupy_Customized_Packet.PCCP(*[_to(dev, pair[k]) for k in ("REF", "SCI", "mREF", "mSCI")], FC, w,
                                                KerPolyOrder=DK, BGPolyOrder=DB, ConstPhotRatio=CPR,
                                                CUDA_DEVICE_4SUBTRACT=str(dev.index), VERBOSE_LEVEL=0)
    assert rel_rms_err(diff.cpu().numpy(), diff_o) <= 1e-6
    # apply-only with the oracle's solution: tight gate
    from sfft_amd.plan import get_plan
    plan = get_plan(N0, N1, w, DK, DB, CPR, dev.index)
    if FC == "REF":
        I, J = pair["REF"], pair["SCI"]
    else:
        I, J = pair["SCI"], pair["REF"]
    d2 = plan.apply(_to(dev, I), _to(dev, J), _to(dev, sol_o)).cpu().numpy()
    p = O.SSC(N0, N1, w, DK, DB, CPR)
    d2_o = O.ESS(I, J, p, SFFTSolution=sol_o, Subtract=True, workers=8)[1]
    assert rms(d2 - d2_o) <= 1e-10 * rms(J)


def test_same_tensor_as_its_own_mask_reuses_spectra(dev):
    """GSS with the full images passed as their own masks ('same', SFFTSubtract.py:849): the C ABI then skips the
    second set of forward transforms; the result must equal the run with separate (identical) buffers."""
    from sfft_amd.plan import get_plan
    from sfft_amd.utils.synthetic import make_pair
    N0, N1 = 256, 512
    pair = make_pair(N0, N1, seed=5, mask=False)
    plan = get_plan(N0, N1, 3, 2, 1, True, dev.index)
    R, S = _to(dev, pair["REF"]), _to(dev, pair["SCI"])
    sol_a, diff_a = plan.subtract(R, S, R, S)
    sol_b, diff_b = plan.subtract(R, S, R.clone(), S.clone())
    assert torch.equal(sol_a, sol_b)
    assert rms((diff_a - diff_b).cpu().numpy()) <= 1e-12 * rms(pair["SCI"])


CONTAM_NAMES = [n for n in NAMES if "ContamMask_I" in np.load(os.path.join(os.path.dirname(__file__), "golden", n + ".npz")).files]


@pytest.mark.parametrize("name", CONTAM_NAMES)
def test_contamination_mask_matches_reference(dev, name):
    """ContamMask_I -> ContamMask_CI through GSS (SFFTSubtract.py:907-921) against the reference's own GSS output.  The fixture
    also holds the convolved mask the reference thresholds (the DIFF of its third ESS call), so the comparison can be exact:
    (a) with the reference's Solution the convolved mask agrees to 1e-10 of its maximum and the boolean mask is identical on
    every pixel that is not within 1e-9 of the threshold; (b) end to end (own solution, cond up to 4e13) the mask is identical
    on every pixel farther than 1e-6 from the threshold."""
    from sfft_amd.sfftcore import SingleSFFTConfigure, GeneralSFFTSubtract
    g = load_golden(name)
    m = g["meta"]
    assert m["ForceConv"] == "REF"
    I, J, mI, mJ, _ = packet_roles(g)
    D_ref, cm_ref = g["ContamD"], g["ContamMask_CI"]
    assert np.array_equal(cm_ref, D_ref < -0.001) and cm_ref.any() and not cm_ref.all()
    cfg = SingleSFFTConfigure.SSC(m["N0"], m["N1"], m["KerHW"], m["DK"], m["DB"], bool(m["CPR"]), VERBOSE_LEVEL=0,
                                  CUDA_DEVICE_4SUBTRACT=dev.index)
    # (a) kernel-only reference solution applied to the mask with J = 0
    tsol = g["Solution"].copy()
    tsol[-cfg[0]["Fpq"]:] = 0.0
    D = cfg[1]["plan"].apply(_to(dev, g["ContamMask_I"].astype(np.float64)), _to(dev, np.zeros_like(J)), _to(dev, tsol)).cpu().numpy()
    assert np.max(np.abs(D - D_ref)) <= 1e-10 * np.max(np.abs(D_ref))
    clear = np.abs(D_ref + 0.001) >= 1e-9
    assert np.array_equal((D < -0.001)[clear], cm_ref[clear]) and clear.mean() > 0.99
    # (b) the operator itself
    sol, diff, cmask = GeneralSFFTSubtract.GSS(I, J, mI, mJ, cfg, ContamMask_I=g["ContamMask_I"], VERBOSE_LEVEL=0)
    assert cmask.dtype == bool and cmask.shape == cm_ref.shape
    assert rel_rms_err(diff, g["DIFF"]) <= 1e-6
    clear = np.abs(D_ref + 0.001) >= 1e-6 * max(1.0, float(np.max(np.abs(D_ref))))
    assert np.array_equal(cmask[clear], cm_ref[clear]) and clear.mean() > 0.98


def test_contamination_mask_matches_oracle(dev):
    from oracle import sfft_oracle as O
    from sfft_amd.sfftcore import SingleSFFTConfigure, GeneralSFFTSubtract
    from sfft_amd.utils.synthetic import make_pair
    N0, N1, w = 128, 96, 3
    pair = make_pair(N0, N1, seed=99, mask=False)
    cm = np.zeros((N0, N1), dtype=bool)
    cm[40:44, 50:53] = True
    cm[100, 10] = True
    cfg = SingleSFFTConfigure.SSC(N0, N1, w, 1, 1, True, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index)
    sol, diff, cmask = GeneralSFFTSubtract.GSS(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], cfg,
                                               ContamMask_I=cm, VERBOSE_LEVEL=0)
    p = O.SSC(N0, N1, w, 1, 1, True)
    sol_o, diff_o, cmask_o = O.GSS(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], p, ContamMask_I=cm)
    assert rel_rms_err(diff, diff_o) <= 1e-6
    assert cmask.dtype == bool and cmask.shape == (N0, N1)
    # exact away from the threshold: the oracle's convolved mask tells which pixels sit on it
    tsol = sol_o.copy()
    tsol[-p["Fpq"]:] = 0.0
    D_o = O.ESS(cm.astype(np.float64), np.zeros((N0, N1)), p, tsol, True)[1]
    clear = np.abs(D_o + 0.001) >= 1e-6
    assert np.array_equal(cmask[clear], cmask_o[clear]) and clear.mean() > 0.98
    assert cmask[41, 51] and cmask[100, 10]


def test_error_behaviour(dev):
    from sfft_amd.sfftcore import SingleSFFTConfigure, ElementalSFFTSubtract, GeneralSFFTSubtract
    from sfft_amd import PureCupy_Customized_Packet
    cfg = SingleSFFTConfigure.SSC(64, 64, 2, 1, 1, True, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index)
    assert cfg[0]["NEQ"] == 3 * 25 + 3 and cfg[0]["NEQ_FSfree"] == 3 * 25 + 3 - 2 and cfg[0]["Fijab"] == 75
    a = np.zeros((64, 32))
    with pytest.raises(Exception, match=r"INCONSISTENT shape of input images I & J, \[64, 64\] required!"):
        ElementalSFFTSubtract.ESS(a, a, cfg, VERBOSE_LEVEL=0)
    b = np.zeros((64, 64))
    with pytest.raises(Exception, match="Input images should have same size!"):
        GeneralSFFTSubtract.GSS(b, b, a, a, cfg, VERBOSE_LEVEL=0)
    bad = np.ones((64, 64)); bad[3, 3] = np.nan
    t = lambda x: _to(dev, x)
    with pytest.raises(AssertionError, match="masked reference image contains NaNs"):
        PureCupy_Customized_Packet.PCCP(t(b), t(b), t(bad), t(b), "REF", 2, VERBOSE_LEVEL=0)
    with pytest.raises(AssertionError):
        PureCupy_Customized_Packet.PCCP(t(b), t(b), t(b), t(b), "BOTH", 2, VERBOSE_LEVEL=0)
    with pytest.raises(AssertionError, match="dtype"):
        PureCupy_Customized_Packet.PCCP(t(b).float(), t(b), t(b), t(b), "REF", 2, VERBOSE_LEVEL=0)
    # exactly singular system (all-zero masked pair): Cholesky fails, LU meets a zero pivot -> LinAlgError like numpy
    with pytest.raises(np.linalg.LinAlgError):
        ElementalSFFTSubtract.ESS(b, b, cfg, VERBOSE_LEVEL=0)
    # ... and a failed pair must not poison the handle: the same config / plan then solves a good pair, through
    # GSS (deferred status check) and through ESS, to the same result as a fresh plan
    from sfft_amd.plan import Plan
    from sfft_amd.utils.synthetic import make_pair
    good = make_pair(64, 64, seed=3, mask=False)
    with pytest.raises(np.linalg.LinAlgError):
        GeneralSFFTSubtract.GSS(b, b, b, b, cfg, VERBOSE_LEVEL=0)
    sol, diff, _ = GeneralSFFTSubtract.GSS(good["REF"], good["SCI"], good["mREF"], good["mSCI"], cfg, VERBOSE_LEVEL=0)
    fresh = Plan(64, 64, 2, 1, 1, True, device=dev.index)
    sol_f, diff_f = fresh.subtract(t(good["REF"]), t(good["SCI"]), t(good["mREF"]), t(good["mSCI"]))
    assert np.array_equal(sol, sol_f.cpu().numpy()) and np.array_equal(diff, diff_f.cpu().numpy())
    assert cfg[1]["plan"].query("LAST_SOLVER") == 1 and np.isfinite(diff).all()
    # a prime side above the on-chip Bluestein limit is a supported size since round 5 (Bluestein through the four-step transform): the
    # reference takes any size (cuFFT / numpy.fft, SFFTSubtract.py:153-154); only sides above 16384 with no on-chip factorisation are refused
    cfg_p = SingleSFFTConfigure.SSC(10007, 64, 2, 1, 1, True, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index)
    assert cfg_p[0]["N0"] == 10007
    with pytest.raises(Exception, match="not supported by this build"):
        SingleSFFTConfigure.SSC(16411, 64, 2, 1, 1, True, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index)   # prime > 16384


# ------------------------------------------------------------------------------------------------
# (c) BASELINE size: properties that need no oracle
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big(dev):
    from sfft_amd.plan import get_plan
    from sfft_amd.utils.synthetic import make_pair
    N = 4096
    pair = make_pair(N, N, seed=1234, mask=True, sky=0.0, bkg_scale=0.05)
    plan = get_plan(N, N, 8, 2, 2, True, dev.index)
    g = {k: _to(dev, v) for k, v in pair.items()}
    return plan, pair, g


def test_baseline_size_identity_kernel_and_background(dev, big):
    """Solution = unit delta kernel (a_00,centre = N0*N1, SFFTSolutionReader.py:63-64) + background b_pq:
    DIFF must equal J - I - sum_pq b_pq cx^p cy^q to rounding."""
    plan, pair, g = big
    N = plan.N0
    sol = np.zeros(plan.NEQ)
    Fab, cen = 17 * 17, 8 * 17 + 8
    sol[cen] = float(N * N)
    b = np.array([3.0, -1.0, 0.5, 2.0, 0.25, -0.75])        # (0,0),(0,1),(0,2),(1,0),(1,1),(2,0)
    sol[plan.Fijab:] = b
    D = plan.apply(g["REF"], g["SCI"], _to(dev, sol)).cpu().numpy()
    cx = ((np.arange(N) + 1.0) / N)[:, None]
    cy = ((np.arange(N) + 1.0) / N)[None, :]
    B = b[0] + b[1] * cy + b[2] * cy ** 2 + b[3] * cx + b[4] * cx * cy + b[5] * cx ** 2
    expect = pair["SCI"] - pair["REF"] - B
    assert rms(D - expect) <= 1e-10 * rms(pair["SCI"])


def test_baseline_size_one_tensor_as_both_images(dev, big):
    """solve(I, I) on the 4096^2 path: the fused row pass writes one row-moment set per distinct source image, so a call whose
    two images are ONE buffer must fall back to the separate moment launches (round-2 advice).  The system of (I, I) must equal
    the system of (I, copy of I) and the solution must be the identity kernel with zero background."""
    plan, pair, g = big
    I = g["mREF"]
    sol_a = plan.solve(I, I)
    LH_a, rhs_a = plan.get_system()
    sol_b = plan.solve(I, I.clone())
    LH_b, rhs_b = plan.get_system()
    LH_a, LH_b, rhs_a, rhs_b = LH_a.cpu().numpy(), LH_b.cpu().numpy(), rhs_a.cpu().numpy(), rhs_b.cpu().numpy()
    nk = plan.Fijab                       # the two moment routes sum in different orders: equal block by block to the usual 1e-11
    for blk in (np.s_[:nk, :nk], np.s_[:nk, nk:], np.s_[nk:, nk:]):
        assert np.abs(LH_a[blk] - LH_b[blk]).max() <= 1e-11 * np.abs(LH_b[blk]).max()
    for blk in (np.s_[:nk], np.s_[nk:]):
        assert np.abs(rhs_a[blk] - rhs_b[blk]).max() <= 1e-11 * np.abs(rhs_b[blk]).max()
    s = sol_a.cpu().numpy()
    assert np.abs(s - sol_b.cpu().numpy()).max() <= 1e-5 * plan.N0 * plan.N1
    N = plan.N0
    expect = np.zeros(plan.NEQ)
    expect[8 * 17 + 8] = float(N * N)
    assert np.abs(s - expect).max() <= 1e-5 * N * N


def test_baseline_size_shift_kernel_and_linearity(dev, big):
    """A pure shift kernel (delta at (a,b)) reproduces a circularly shifted I; apply is linear in the solution."""
    plan, pair, g = big
    N = plan.N0
    Fab, L, w = 289, 17, 8
    a, b = 3, -5
    sol1 = np.zeros(plan.NEQ)
    ab = (a + w) * L + (b + w)
    sol1[ab] = float(N * N)          # modified delta basis: K_ab = delta(a,b) - delta(0,0); centre coefficient = kernel sum
    sol1[w * L + w] = float(N * N)
    D1 = plan.apply(g["REF"], g["SCI"], _to(dev, sol1)).cpu().numpy()
    shifted = np.roll(pair["REF"], shift=(a, b), axis=(0, 1))
    assert rms(D1 - (pair["SCI"] - shifted)) <= 1e-10 * rms(pair["SCI"])
    rng = np.random.default_rng(0)
    sol2 = rng.normal(size=plan.NEQ) * 1e5
    D2 = plan.apply(g["REF"], g["SCI"], _to(dev, sol2)).cpu().numpy()
    D12 = plan.apply(g["REF"], g["SCI"], _to(dev, sol1 + sol2)).cpu().numpy()
    J = pair["SCI"]
    # J - D is linear in the solution
    assert rms((J - D12) - ((J - D1) + (J - D2))) <= 1e-10 * rms(J - D2)


def test_baseline_size_system_is_symmetric_gram_and_solution_is_stationary(dev, big):
    plan, pair, g = big
    sol, diff = plan.subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"])
    assert plan.query("LAST_SOLVER") in (1, 2)
    LH, rhs = plan.get_system()
    asym = float((LH - LH.T).abs().max() / LH.abs().max())
    assert asym <= 1e-13
    # residual of the normal equations on the stripe-free system
    idx = np.ones(plan.NEQ, dtype=bool)
    idx[[ij * 289 + 8 * 17 + 8 for ij in range(1, 6)]] = False
    idx_t = torch.from_numpy(np.where(idx)[0]).to(dev)
    A = LH[idx_t][:, idx_t]
    r = A @ sol[idx_t] - rhs[idx_t]
    assert float(r.abs().max()) <= 1e-7 * float(rhs.abs().max())
    d = diff.cpu().numpy()
    assert np.isfinite(d).all()
    # the fit removes the stars: residual RMS is of the order of the noise (3 in REF, blurred + 3 in SCI)
    assert rms(d) < 8.0
    # recovered photometric ratio: kernel sum = a_00,centre / (N0*N1) (SFFTSolutionReader.py:173-181) ~ 1.3
    ksum = float(sol[8 * 17 + 8].item()) / (4096.0 * 4096.0)
    assert abs(ksum - 1.3) < 0.05


# ------------------------------------------------------------------------------------------------
# (d) the large BASELINE shapes (configs 3 and 5 sizes; four-step transforms): properties only
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [(6144, 6144, 8, 2, 2), (9232, 9216, 12, 3, 3)])
def test_large_shapes_properties(dev, cfg):
    from sfft_amd.plan import get_plan, clear_plan_cache
    clear_plan_cache()
    N0, N1, w, DK, DB = cfg
    rng = np.random.default_rng(N0)
    # cheap synthetic pair: smooth blobs + noise (the star renderer is too slow for 85 Mpix in a test)
    x = np.linspace(0, 40 * np.pi, N0)[:, None]
    y = np.linspace(0, 36 * np.pi, N1)[None, :]
    base = 50.0 * (np.sin(x) * np.cos(y)) ** 8
    REF = base + rng.normal(0, 1.0, (N0, N1))
    SCI = 1.2 * (0.6 * base + 0.2 * np.roll(base, 1, 0) + 0.2 * np.roll(base, -1, 1)) + 3.0 + rng.normal(0, 1.0, (N0, N1))
    plan = get_plan(N0, N1, w, DK, DB, True, dev.index)
    R, S = _to(dev, REF), _to(dev, SCI)
    L = 2 * w + 1
    Fij = (DK + 1) * (DK + 2) // 2
    # (1) identity kernel + constant background: DIFF = SCI - REF - b00 exactly
    sol = np.zeros(plan.NEQ)
    sol[w * L + w] = float(N0) * float(N1)
    sol[plan.Fijab] = 2.5
    D = plan.apply(R, S, _to(dev, sol)).cpu().numpy()
    assert rms(D - (SCI - REF - 2.5)) <= 1e-10 * rms(SCI)
    # (2) shift kernel reproduces a circular shift
    a, b = -w, w
    sol = np.zeros(plan.NEQ)
    sol[(a + w) * L + (b + w)] = float(N0) * float(N1)
    sol[w * L + w] = float(N0) * float(N1)
    D = plan.apply(R, S, _to(dev, sol)).cpu().numpy()
    assert rms(D - (SCI - np.roll(REF, (a, b), (0, 1)))) <= 1e-10 * rms(SCI)
    del D
    # (3) end to end: symmetric Gram system, stationary solution, sane residual and photometric ratio
    solution, diff = plan.subtract(R, S, R, S)
    LH, rhs = plan.get_system()
    assert float((LH - LH.T).abs().max() / LH.abs().max()) <= 1e-13
    idx = np.ones(plan.NEQ, dtype=bool)
    idx[[ij * L * L + w * L + w for ij in range(1, Fij)]] = False
    it = torch.from_numpy(np.where(idx)[0]).to(dev)
    r = LH[it][:, it] @ solution[it] - rhs[it]
    assert float(r.abs().max()) <= 1e-6 * float(rhs.abs().max())
    d = diff.cpu().numpy()
    assert np.isfinite(d).all() and rms(d) < 2.0
    assert abs(float(solution[w * L + w].item()) / (float(N0) * float(N1)) - 1.2) < 0.05
    clear_plan_cache()


# ------------------------------------------------------------------------------------------------
# (d2) BASELINE config 3: B-spline kernel, degree 2 with 2 x 2 internal knots (Fij = 25, NEQ 7231, 7207 solved), constant
# scaling, polynomial background of degree 2, KerHW 8 -- at 6144 x 6144 by properties, and on a 6144 x 96 strip (same system
# size: the 256-column outer-blocked Cholesky with the rank-256 chol_syrk update, 325 Omega passes) against the oracle.
# ------------------------------------------------------------------------------------------------
def _config3(N0, N1, dev):
    from sfft_amd.BSplineSFFT import SingleSFFTConfigure as BSSC
    kx, ky = [N0 / 3 + 0.5, 2 * N0 / 3 + 0.5], [N1 / 3 + 0.5, 2 * N1 / 3 + 0.5]
    cfg = BSSC.SSC(NX=N0, NY=N1, KerHW=8, KerSpType="B-Spline", KerSpDegree=2, KerIntKnotX=kx, KerIntKnotY=ky,
                   SEPARATE_SCALING=True, ScaSpDegree=0, BkgSpType="Polynomial", BkgSpDegree=2, VERBOSE_LEVEL=0,
                   CUDA_DEVICE_4SUBTRACT=dev.index)
    assert (cfg[0]["Fij"], cfg[0]["NEQ"], cfg[0]["NEQt"], cfg[0]["SCALING_MODE"]) == (25, 7231, 7207, "SEPARATE-CONSTANT")
    return cfg, kx, ky


def _blob_pair(N0, N1, seed, ratio=1.25, sky=2.0):
    rng = np.random.default_rng(seed)
    x = np.linspace(0, N0 / 100.0 * np.pi, N0)[:, None]
    y = np.linspace(0, max(N1 / 110.0, 1.0) * np.pi, N1)[None, :]
    base = 80.0 * (np.sin(x) * np.cos(y)) ** 8
    REF = base + rng.normal(0, 1.0, (N0, N1))
    SCI = ratio * (0.6 * base + 0.2 * np.roll(base, 1, 0) + 0.2 * np.roll(base, -1, 1)) + sky + rng.normal(0, 1.0, (N0, N1))
    keep = base > 0.5
    return REF, SCI, np.where(keep, REF, 0.0), np.where(keep, SCI, 0.0)


def _tied_residual(LH, rhs, sol, ij00):
    """Residual of the tied system P^T A P x_t = P^T b (TweakLS, BSplineSFFT.py:2201-2272) at the restored solution x = P x_t."""
    r = LH @ sol - rhs
    tied = r[ij00].sum()
    keep = np.ones(r.shape[0], dtype=bool)
    keep[ij00] = False
    return max(float(np.max(np.abs(r[keep]))), abs(float(tied)))


def test_config3_bspline_6144(dev):
    from sfft_amd.plan import clear_plan_cache
    from sfft_amd.BSplineSFFT import GeneralSFFTSubtract_PureCupy as BGSSPC
    clear_plan_cache()
    N = 6144
    cfg, _, _ = _config3(N, N, dev)
    plan = cfg[1]["plan"]
    REF, SCI, mREF, mSCI = _blob_pair(N, N, 3, sky=0.0)   # sky-subtracted frames: a sky step at the mask edge is not in the model
    R, S = _to(dev, REF), _to(dev, SCI)
    w, L, Fab, Fij = 8, 17, 289, 25
    ij00 = np.arange(w * L + w, Fij * Fab, Fab)
    PN = float(N) * float(N)
    # (1) the B-spline factors are a partition of unity: equal centre coefficients are the identity kernel
    sol = np.zeros(plan.NEQ)
    sol[ij00] = PN
    sol[plan.Fijab] = 2.5
    D = plan.apply(R, S, _to(dev, sol)).cpu().numpy()
    assert rms(D - (SCI - REF - 2.5)) <= 1e-10 * rms(SCI)
    # (2) the same for a shift kernel
    a, b = 5, -8
    sol = np.zeros(plan.NEQ)
    sol[ij00] = PN
    sol[ij00 - (w * L + w) + (a + w) * L + (b + w)] = PN
    D = plan.apply(R, S, _to(dev, sol)).cpu().numpy()
    assert rms(D - (SCI - np.roll(REF, (a, b), (0, 1)))) <= 1e-10 * rms(SCI)
    del D
    # (3) end to end on a masked pair that differs from the full pair (the apply pass's transforms run on the second stream)
    solution, diff, _ = BGSSPC.GSS(R, S, _to(dev, mREF), _to(dev, mSCI), cfg, VERBOSE_LEVEL=0)
    assert plan.query("LAST_SOLVER") == 1
    LH, rhs = plan.get_system()
    assert float((LH - LH.T).abs().max() / LH.abs().max()) <= 1e-13
    sol_h = solution.cpu().numpy()
    assert np.all(sol_h[ij00] == sol_h[ij00[0]])                     # tied scaling
    res = _tied_residual(LH.cpu().numpy(), rhs.cpu().numpy(), sol_h, ij00)
    assert res <= 1e-6 * float(rhs.abs().max())
    del LH
    d = diff.cpu().numpy()
    assert np.isfinite(d).all() and rms(d) < 2.0
    assert abs(sol_h[ij00[0]] / PN - 1.25) < 0.05
    # apply is deterministic and linear in the solution at this size too
    D1 = plan.apply(R, S, solution).cpu().numpy()
    assert np.array_equal(D1, d)
    clear_plan_cache()


@pytest.mark.parametrize("shape", [(6144, 96), (96, 6144), (768, 768), (1536, 1536)], ids=["cols6144", "rows6144", "square768", "square1536"])
def test_config3_bspline_strip_matches_oracle(dev, shape):
    """6144 x 96 strip with config 3's basis: the same 7231-unknown system as the full frame (outer-blocked Cholesky, 325 Omega
    passes, tied scaling), small enough for the oracle: LHMAT / RHb <= 1e-11, apply-only DIFF <= 1e-10 RMS(J), end to end <= 1e-6.
    The transposed strip (96 x 6144) runs the full-width 6144-point ROW passes (forward r2c of the B-spline stage planes, inverse c2r
    with the DIFF epilogue) with the config's own basis tables.  The 768 x 768 and 1536 x 1536 squares (round 4) are sizes at which the GROUP SCHEDULING
    of the 25-plane Omega launch matters -- 25 (49) column tiles x 113 pass groups x several row chunks in one grid, as at 6144^2 -- and are
    still within the oracle's reach."""
    from oracle import bspline_oracle as BO
    from sfft_amd.plan import clear_plan_cache
    from sfft_amd.BSplineSFFT import GeneralSFFTSubtract as BGSS, ElementalSFFTSubtract as BESS
    clear_plan_cache()
    N0, N1 = shape
    cfg, kx, ky = _config3(N0, N1, dev)
    plan = cfg[1]["plan"]
    REF, SCI, mREF, mSCI = _blob_pair(N0, N1, 4)
    basis = BO.make_basis(N0, N1, "B-Spline", 2, kx, ky, "Polynomial", 2)
    p = BO.SSC(N0, N1, 8, basis, True)
    assert p["NEQ"] == 7231
    ncpu = min(32, os.cpu_count() or 1)
    LH_o, rhs_o = BO.establish_system(mREF, mSCI, p, basis, workers=ncpu)
    sol, D, _ = BGSS.GSS(REF, SCI, mREF, mSCI, cfg, VERBOSE_LEVEL=0)
    assert plan.query("LAST_SOLVER") == 1
    LH, rhs = plan.get_system()
    assert np.max(np.abs(LH.cpu().numpy() - LH_o)) <= 1e-11 * np.max(np.abs(LH_o))
    assert np.max(np.abs(rhs.cpu().numpy() - rhs_o)) <= 1e-11 * np.max(np.abs(rhs_o))
    del LH
    sol_o = BO.solve_system(LH_o, rhs_o, p)
    D_o = BO.subtract(REF, SCI, sol_o, p, basis, workers=ncpu)
    Da = BESS.ESS(REF, SCI, cfg, SFFTSolution=sol_o, Subtract=True, VERBOSE_LEVEL=0)[1]
    assert rms(Da - D_o) <= 1e-10 * rms(SCI)
    assert rel_rms_err(D, D_o) <= 1e-6
    ij00 = np.arange(8 * 17 + 8, 25 * 289, 289)
    assert np.all(sol[ij00] == sol[ij00[0]])
    if shape == (6144, 96):     # the same system through the reference's own solver, LU with partial pivoting (lu.hpp), at n = 7207: same gates
        plan.set_force_lu(True)
        try:
            sol_lu, D_lu, _ = BGSS.GSS(REF, SCI, mREF, mSCI, cfg, VERBOSE_LEVEL=0)
            assert plan.query("LAST_SOLVER") == 2
        finally:
            plan.set_force_lu(False)
        assert rel_rms_err(D_lu, D_o) <= 1e-6
        assert np.all(sol_lu[ij00] == sol_lu[ij00[0]])
        assert np.max(np.abs(sol_lu - sol)) <= 1e-6 * np.max(np.abs(sol))
    clear_plan_cache()


# ------------------------------------------------------------------------------------------------
# (d3) BASELINE config 2 at FULL size against the oracle: 4096 x 4096, KerHW 8, orders 2/2.  The oracle (pinned by the
# reference-made fixtures, including two at this kernel geometry) builds the 1740 x 1740 system and the difference image
# on the host cores in a few minutes; every Omega / Gamma / Theta lag patch, the Phi / Delta closed forms and the
# mixed-domain apply are compared element by element at the size the benchmark runs.
# ------------------------------------------------------------------------------------------------
def test_config2_full_size_matches_oracle(dev, big):
    from oracle import sfft_oracle as O
    plan, pair, g = big
    N = plan.N0
    ncpu = min(64, os.cpu_count() or 1)
    p = O.SSC(N, N, 8, 2, 2, True)
    LH_o, rhs_o = O.establish_system(pair["mREF"], pair["mSCI"], p, workers=ncpu)
    sol, diff = plan.subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"])
    LH, rhs = plan.get_system()
    LH, rhs = LH.cpu().numpy(), rhs.cpu().numpy()
    assert np.max(np.abs(LH - LH_o)) <= 1e-11 * np.max(np.abs(LH_o))
    assert np.max(np.abs(rhs - rhs_o)) <= 1e-11 * np.max(np.abs(rhs_o))
    nk = p["Fijab"]      # block by block: the kernel block is far smaller than the background block
    for blk, blk_o in ((LH[:nk, :nk], LH_o[:nk, :nk]), (LH[:nk, nk:], LH_o[:nk, nk:]), (LH[nk:, :nk], LH_o[nk:, :nk]),
                       (LH[nk:, nk:], LH_o[nk:, nk:]), (rhs[:nk], rhs_o[:nk]), (rhs[nk:], rhs_o[nk:])):
        assert np.max(np.abs(blk - blk_o)) <= 1e-11 * np.max(np.abs(blk_o))
    # apply-only with the oracle's own solution, then end to end
    sol_o = O.solve_system(LH_o, rhs_o, p)
    D_o = O.ESS(pair["REF"], pair["SCI"], p, sol_o, True, ncpu)[1]
    Da = plan.apply(g["REF"], g["SCI"], _to(dev, sol_o)).cpu().numpy()
    assert rms(Da - D_o) <= 1e-10 * rms(pair["SCI"])
    assert rel_rms_err(diff.cpu().numpy(), D_o) <= 1e-6
    # the same pair through the reference's own solver, LU with partial pivoting (lu.hpp; n = 1735): same end-to-end gate
    plan.set_force_lu(True)
    try:
        sol_lu, diff_lu = plan.subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"])
        assert plan.query("LAST_SOLVER") == 2
    finally:
        plan.set_force_lu(False)
    assert rel_rms_err(diff_lu.cpu().numpy(), D_o) <= 1e-6
    # (the two factorisations of this ill-conditioned Gram matrix agree to cond x eps: 1.4e-7 of the largest coefficient)
    assert float((sol_lu - sol).abs().max()) <= 1e-6 * float(sol.abs().max())


@pytest.mark.parametrize("shape", [(9232, 128), (128, 9216)], ids=["cols9232", "rows9216"])
def test_config5_strip_matches_oracle(dev, shape):
    """(The 128 x 9216 strip: config 5's full-width 9216-point ROW passes with KerHW 12 and orders 3 / 3.)
    BASELINE config 5's geometry on a 9232 x 128 strip: the four-step 9232-point axis (16 x 577), KerHW 12 (49 x 49 Omega lag patches: the
    decimated vector kernel in two lag bands), orders 3 / 3 (NEQ 6260: the outer-blocked Cholesky), vconv_mixed<3,12> -- against the oracle:
    LHMAT / RHb <= 1e-11 block by block, apply-only DIFF <= 1e-10 RMS(J), end to end <= 1e-6."""
    from oracle import sfft_oracle as O
    from sfft_amd.plan import Plan
    (N0, N1), w = shape, 12
    if N0 < N1:
        # (the sin x cos y blob field is numerically rank deficient on a wide strip: cond(LHMAT) = 8e17 at 128 x 9216, where LU and Cholesky on
        #  the ORACLE's own matrix already differ by 4e-6 in DIFF; a seeded star field is not)
        from sfft_amd.utils.synthetic import make_pair
        pr = make_pair(N0, N1, seed=11, mask=True, density=400.0)
        REF, SCI, mREF, mSCI = pr["REF"], pr["SCI"], pr["mREF"], pr["mSCI"]
    else:
        REF, SCI, mREF, mSCI = _blob_pair(N0, N1, 5)
    plan = Plan(N0, N1, w, 3, 3, True, device=dev.index)
    assert plan.NEQ == 6260
    ncpu = min(64, os.cpu_count() or 1)
    p = O.SSC(N0, N1, w, 3, 3, True)
    LH_o, rhs_o = O.establish_system(mREF, mSCI, p, workers=ncpu)
    R, S, mR, mS = _to(dev, REF), _to(dev, SCI), _to(dev, mREF), _to(dev, mSCI)
    sol, diff = plan.subtract(R, S, mR, mS)
    assert plan.query("LAST_SOLVER") == 1
    LH, rhs = plan.get_system()
    LH, rhs = LH.cpu().numpy(), rhs.cpu().numpy()
    nk = p["Fijab"]
    for blk, blk_o in ((LH[:nk, :nk], LH_o[:nk, :nk]), (LH[:nk, nk:], LH_o[:nk, nk:]), (LH[nk:, nk:], LH_o[nk:, nk:]),
                       (rhs[:nk], rhs_o[:nk]), (rhs[nk:], rhs_o[nk:])):
        assert np.max(np.abs(blk - blk_o)) <= 1e-11 * np.max(np.abs(blk_o))
    del LH
    sol_o = O.solve_system(LH_o, rhs_o, p)
    D_o = O.ESS(REF, SCI, p, sol_o, True, ncpu)[1]
    Da = plan.apply(R, S, _to(dev, sol_o)).cpu().numpy()
    assert rms(Da - D_o) <= 1e-10 * rms(SCI)
    assert rel_rms_err(diff.cpu().numpy(), D_o) <= 1e-6
    # the same system (n = 6251 after the stripes are removed) through the reference's own solver semantics -- LU with partial pivoting
    # (lu.hpp; SFFTSubtract.py:15-23, 398-403): same gate on DIFF against the oracle, whose solve_system IS numpy's LU.  (bench.py's
    # `diff_max_rel_vs_cholesky` of 2e-5 at config 5 is a MAX over pixels between two backward-stable solutions of a system with
    # cond ~ 1e13+ on blob data; the pixel-RMS distance to the oracle is what the tolerance of north_star is stated in.)
    plan.set_force_lu(True)
    try:
        sol_lu, diff_lu = plan.subtract(R, S, mR, mS)
        assert plan.query("LAST_SOLVER") == 2
    finally:
        plan.set_force_lu(False)
    assert rel_rms_err(diff_lu.cpu().numpy(), D_o) <= 1e-6
    plan.close()


# ------------------------------------------------------------------------------------------------
# (e) B-spline form (sfft_amd.BSplineSFFT): golden vectors from the reference's dev-version Numpy backend
# ------------------------------------------------------------------------------------------------
from _golden import bspline_golden_names, load_bspline_golden

BS_NAMES = bspline_golden_names()


def _bs_config(m, dev):
    from sfft_amd.BSplineSFFT import SingleSFFTConfigure as BSSC
    return BSSC.SSC(NX=m["N0"], NY=m["N1"], KerHW=m["w"], KerSpType=m["KerSpType"], KerSpDegree=m["KerSpDegree"],
                    KerIntKnotX=m["KerIntKnotX"], KerIntKnotY=m["KerIntKnotY"], SEPARATE_SCALING=bool(m["CPR"]), ScaSpDegree=0,
                    BkgSpType=m["BkgSpType"], BkgSpDegree=m["BkgSpDegree"], BkgIntKnotX=m["BkgIntKnotX"],
                    BkgIntKnotY=m["BkgIntKnotY"], VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index)


@pytest.mark.parametrize("name", BS_NAMES)
def test_bspline_matches_reference(dev, name):
    from sfft_amd.BSplineSFFT import GeneralSFFTSubtract as BGSS, ElementalSFFTSubtract as BESS
    g = load_bspline_golden(name)
    m = g["meta"]
    cfg = _bs_config(m, dev)
    assert (cfg[0]["NEQ"], cfg[0]["Fij"], cfg[0]["Fpq"]) == (m["NEQ"], m["Fij"], m["Fpq"])
    plan = cfg[1]["plan"]
    plan.solve(_to(dev, g["mREF"]), _to(dev, g["mSCI"]))
    LH, rhs = plan.get_system()
    assert np.max(np.abs(LH.cpu().numpy() - g["LHMAT"])) <= 1e-11 * np.max(np.abs(g["LHMAT"]))
    assert np.max(np.abs(rhs.cpu().numpy() - g["RHb"])) <= 1e-11 * np.max(np.abs(g["RHb"]))
    # the system the factorisation received: TweakLS of the reference (BSplineSFFT.py:2170-2338) -- polynomial kernels drop the
    # ij00[1:] unknowns, B-spline kernels tie them to ij00[0] (rows and columns summed): P^T LHMAT P, P^T RHb
    A, b, idx = (t.cpu().numpy() for t in plan.get_solver_system())
    NEQ, Lk = m["NEQ"], 2 * m["w"] + 1
    ij00_ = np.arange(m["w"] * Lk + m["w"], m["Fij"] * Lk * Lk, Lk * Lk)
    keep = np.setdiff1d(np.arange(NEQ), ij00_[1:]) if bool(m["CPR"]) else np.arange(NEQ)
    assert np.array_equal(idx, keep)
    P = np.zeros((NEQ, len(keep)))
    P[keep, np.arange(len(keep))] = 1.0
    if bool(m["CPR"]) and m["KerSpType"] == "B-Spline":
        P[ij00_[1:], int(np.where(keep == ij00_[0])[0][0])] = 1.0
    assert np.max(np.abs(A - P.T @ g["LHMAT"] @ P)) <= 1e-11 * np.max(np.abs(g["LHMAT"]))
    assert np.max(np.abs(b - P.T @ g["RHb"])) <= 1e-11 * np.max(np.abs(g["RHb"]))
    D = BESS.ESS(g["REF"], g["SCI"], cfg, SFFTSolution=g["Solution"], Subtract=True, VERBOSE_LEVEL=0)[1]
    assert rms(D - g["DIFF"]) <= 1e-10 * rms(g["SCI"])
    sol, D2, _ = BGSS.GSS(g["REF"], g["SCI"], g["mREF"], g["mSCI"], cfg, VERBOSE_LEVEL=0)
    assert rel_rms_err(D2, g["DIFF"]) <= 1e-6
    if bool(m["CPR"]) and m["KerSpType"] == "B-Spline":
        L = 2 * m["w"] + 1
        ij00 = np.arange(m["w"] * L + m["w"], m["Fij"] * L * L, L * L)
        assert np.all(sol[ij00] == sol[ij00[0]])


def test_bspline_background_matches_oracle(dev):
    """B-spline BACKGROUND variation at a larger shape against the restated oracle (the oracle itself is pinned for B-spline
    backgrounds by the bs_*_bkgbspl* fixtures: the reference's dev-version code with its mistyped function name corrected in memory,
    tests/golden/make_golden_bspline.py; those fixtures run through the parametrised golden tests above)."""
    from oracle import bspline_oracle as BO
    from sfft_amd.BSplineSFFT import SingleSFFTConfigure as BSSC, GeneralSFFTSubtract as BGSS
    from sfft_amd.utils.synthetic import make_pair
    N0, N1, w = 96, 128, 2
    pair = make_pair(N0, N1, seed=77, mask=False, density=500.0)
    kw = dict(KerSpType="B-Spline", KerSpDegree=2, KerIntKnotX=[48.5], KerIntKnotY=[40.5, 88.5],
              BkgSpType="B-Spline", BkgSpDegree=2, BkgIntKnotX=[30.5, 60.5], BkgIntKnotY=[64.5])
    cfg = BSSC.SSC(NX=N0, NY=N1, KerHW=w, SEPARATE_SCALING=True, ScaSpDegree=0, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index, **kw)
    sol, D, _ = BGSS.GSS(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], cfg, VERBOSE_LEVEL=0)
    basis = BO.make_basis(N0, N1, **kw)
    p = BO.SSC(N0, N1, w, basis, True)
    assert cfg[0]["Fij"] == 20 and cfg[0]["Fpq"] == 20 and cfg[0]["NEQ"] == p["NEQ"]
    sol_o, D_o = BO.GSS(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], p, basis, workers=8)
    assert rel_rms_err(D, D_o) <= 1e-6


def test_bspline_packet_fits_and_unsupported_modes(dev, tmp_path):
    from sfft_amd.BSplineSFFT import BSpline_Packet, SingleSFFTConfigure as BSSC
    from sfft_amd.utils import minifits
    g = load_bspline_golden("bs_64x48_w2_bspl1_k2_poly2_const")
    m = g["meta"]
    paths = {}
    for k in ("REF", "SCI", "mREF", "mSCI"):
        paths[k] = str(tmp_path / (k + ".fits"))
        minifits.writeto(paths[k], np.ascontiguousarray(g[k].T))
    fdiff = str(tmp_path / "d.fits")
    sol, diff = BSpline_Packet.BSP(paths["REF"], paths["SCI"], paths["mREF"], paths["mSCI"], FITS_DIFF=fdiff, ForceConv="REF",
                                   GKerHW=m["w"], KerSpType="B-Spline", KerSpDegree=1, KerIntKnotX=m["KerIntKnotX"],
                                   KerIntKnotY=m["KerIntKnotY"], SEPARATE_SCALING=True, ScaSpDegree=0, BkgSpType="Polynomial",
                                   BkgSpDegree=2, CUDA_DEVICE_4SUBTRACT=str(dev.index), VERBOSE_LEVEL=0)
    assert rel_rms_err(diff, g["DIFF"]) <= 1e-6
    h = minifits.header_dict(minifits.getdata(fdiff)[1])
    assert h["KSPTYPE"] == "B-Spline" and h["NKIKX"] == 2 and abs(h["KIKX1"] - 44.5) < 1e-12 and h["SEPSCA"] == "True"
    with pytest.raises(AssertionError):     # polynomial scaling of the kernel's own degree "reduces to ENTANGLED" (BSplineSFFT.py:70-71)
        BSSC.SSC(64, 48, 2, KerSpType="Polynomial", KerSpDegree=1, SEPARATE_SCALING=True, ScaSpDegree=1, VERBOSE_LEVEL=0)
    with pytest.raises(AssertionError):     # REGULARIZE_KERNEL needs XY_REGULARIZE (:88-90)
        BSSC.SSC(64, 48, 2, KerSpType="B-Spline", KerSpDegree=2, REGULARIZE_KERNEL=True, VERBOSE_LEVEL=0)


@pytest.mark.parametrize("shape,w,deg,nk,sep", [((256, 200), 4, 2, 2, True), ((200, 162), 3, 1, 2, False), ((288, 256), 8, 2, 1, True),
                                                ((320, 258), 6, 3, 2, True), ((1536, 1024), 8, 2, 2, True)])
def test_bspline_mixed_domain_apply_equals_fourier_apply(dev, shape, w, deg, nk, sep):
    """B-spline kernels with a full tensor basis (4 x 4 .. 6 x 6 terms, KerHW <= 8) take the mixed-domain apply too (vconv_tensor:
    nky row transforms instead of nkx nky plane transforms, no column transform): same DIFF as the Fourier-domain apply
    (SFFT_NO_VCONV=1) for a random solution vector, and the same end-to-end result."""
    import sfft_amd.BSplineSFFT as B
    from sfft_amd.utils.synthetic import make_pair
    N0, N1 = shape
    pair = make_pair(N0, N1, seed=3 + w, mask=True)
    kx = [N0 * (k + 1) / (nk + 1) + 0.5 for k in range(nk)]
    ky = [N1 * (k + 1) / (nk + 1) + 0.5 for k in range(nk)]
    outs = []
    sol = None
    for no_vconv in (False, True):
        B._PLANS.clear()
        if no_vconv:
            os.environ["SFFT_NO_VCONV"] = "1"
        try:
            cfg = B.SingleSFFTConfigure.SSC(NX=N0, NY=N1, KerHW=w, KerSpType="B-Spline", KerSpDegree=deg, KerIntKnotX=kx, KerIntKnotY=ky,
                                            SEPARATE_SCALING=sep, ScaSpDegree=0, BkgSpType="Polynomial", BkgSpDegree=1, VERBOSE_LEVEL=0,
                                            CUDA_DEVICE_4SUBTRACT=dev.index)
        finally:
            os.environ.pop("SFFT_NO_VCONV", None)
        assert cfg[0]["Fij"] == (deg + 1 + nk) ** 2
        if sol is None:
            rng = np.random.default_rng(17)
            sol = rng.normal(size=cfg[0]["NEQ"])
            sol[:cfg[0]["Fijab"]] *= float(N0) * float(N1) * 0.01
        d_rand = B.ElementalSFFTSubtract.ESS(pair["REF"], pair["SCI"], cfg, SFFTSolution=sol, Subtract=True, VERBOSE_LEVEL=0)[1]
        s2, d2, _ = B.GeneralSFFTSubtract.GSS(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], cfg, VERBOSE_LEVEL=0)
        outs.append((np.asarray(d_rand), np.asarray(s2), np.asarray(d2)))
    B._PLANS.clear()
    (da, sa, ga), (db, sb, gb) = outs
    assert rms(da - db) <= 1e-12 * rms(db)
    assert np.array_equal(sa, sb)
    assert rms(ga - gb) <= 1e-11 * rms(pair["SCI"])


@pytest.mark.parametrize("shape,w,deg,nk", [((256, 200), 4, 2, 2), ((320, 264), 8, 2, 3), ((1024, 2560), 8, 2, 2), ((2304, 512), 6, 1, 3), ((128, 64), 3, 3, 4)])
def test_bspline_sparse_omega_equals_transformed_omega(dev, shape, w, deg, nk):
    """Omega products of B-spline terms whose row (or column) factors have disjoint supports are summed in real space (omega_sparse:
    a few 1-D correlations of image rows / columns, greek.hpp) instead of through the transforms.  Same linear system to rounding and
    the same result as with every product transformed (SFFT_NO_OMG_SPARSE=1); the plan reports how many products took the short
    cut.  Shapes: one step per line, lines of several steps (2560 > OSP_CH = 2048), many knots, a line barely longer than the halo."""
    import sfft_amd.BSplineSFFT as B
    from sfft_amd.utils.synthetic import make_pair
    N0, N1 = shape
    pair = make_pair(N0, N1, seed=11 + w, mask=True)
    kx = [N0 * (k + 1) / (nk + 1) + 0.5 for k in range(nk)]
    ky = [N1 * (k + 1) / (nk + 1) + 0.5 for k in range(nk)]
    outs = []
    for dense in (False, True):
        B._PLANS.clear()
        if dense:
            os.environ["SFFT_NO_OMG_SPARSE"] = "1"
        try:
            cfg = B.SingleSFFTConfigure.SSC(NX=N0, NY=N1, KerHW=w, KerSpType="B-Spline", KerSpDegree=deg, KerIntKnotX=kx, KerIntKnotY=ky,
                                            SEPARATE_SCALING=True, ScaSpDegree=0, BkgSpType="Polynomial", BkgSpDegree=1, VERBOSE_LEVEL=0,
                                            CUDA_DEVICE_4SUBTRACT=dev.index)
            plan = next(iter(B._PLANS.values()))
            nsp = plan.query("OMG_SPARSE")
            try:
                d2 = np.asarray(B.GeneralSFFTSubtract.GSS(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], cfg, VERBOSE_LEVEL=0)[1])
            except np.linalg.LinAlgError:       # (many knots on a small masked image: a term without unmasked support; the systems are still compared)
                d2 = None
                assert nk >= 3
            LH, rhs = plan.get_system()
        finally:
            os.environ.pop("SFFT_NO_OMG_SPARSE", None)
        outs.append((d2, LH.cpu().numpy(), rhs.cpu().numpy(), nsp))
    B._PLANS.clear()
    (da, La, ra, na), (db, Lb, rb, nb) = outs
    assert nb == 0 and na > 0, (na, nb)
    assert np.max(np.abs(La - Lb)) <= 1e-11 * np.max(np.abs(Lb))
    assert np.max(np.abs(ra - rb)) <= 1e-11 * np.max(np.abs(rb))
    assert (da is None) == (db is None)
    if da is not None:
        assert rms(da - db) <= 1e-7 * rms(pair["SCI"])


# ------------------------------------------------------------------------------------------------
# (e2) separately varying scaling + kernel regularisation (BSplineSFFT.py SCALING_MODE 'SEPARATE-VARYING', REGULARIZE_KERNEL).
# The reference has no CPU code for these: the comparison is oracle/bspline_sv_oracle.py (pinned by the reference's NIRCam golden for the
# notebook's configuration, tests/test_nircam_chain.py), itself pinned
# by tests/test_oracle_sv.py (brute-force normal equations, reduction to the golden-pinned ENTANGLED system).
# ------------------------------------------------------------------------------------------------
SV_CASES = [
    dict(name="poly2_sca1", N0=96, N1=80, w=2, ker=("Polynomial", 2, [], []), sca=("Polynomial", 1, [], []), bkg=("Polynomial", 2, [], [])),
    dict(name="bspl2_scabspl1", N0=128, N1=96, w=3, ker=("B-Spline", 2, [64.5], [48.5]), sca=("B-Spline", 1, [], []),
         bkg=("Polynomial", 1, [], [])),
    dict(name="bspl1_scapoly2_full", N0=100, N1=72, w=2, ker=("B-Spline", 1, [50.5], []), sca=("Polynomial", 2, [], []),
         bkg=("B-Spline", 1, [], [36.5])),      # ScaFij == Fij == 6: nothing leaves the system
]


def _sv_setup(c, dev, reg=None):
    from oracle import bspline_oracle as BO, bspline_sv_oracle as SV
    from sfft_amd.BSplineSFFT import SingleSFFTConfigure as BSSC
    from sfft_amd.utils.synthetic import make_pair
    N0, N1, w = c["N0"], c["N1"], c["w"]
    ker, sca, bkg = c["ker"], c["sca"], c["bkg"]
    pair = make_pair(N0, N1, seed=31, mask=True, density=400.0)
    kw = dict(KerSpType=ker[0], KerSpDegree=ker[1], KerIntKnotX=ker[2], KerIntKnotY=ker[3],
              SEPARATE_SCALING=True, ScaSpType=sca[0], ScaSpDegree=sca[1], ScaIntKnotX=sca[2], ScaIntKnotY=sca[3],
              BkgSpType=bkg[0], BkgSpDegree=bkg[1], BkgIntKnotX=bkg[2], BkgIntKnotY=bkg[3])
    if reg is not None:
        kw.update(REGULARIZE_KERNEL=True, **reg)
    cfg = BSSC.SSC(NX=N0, NY=N1, KerHW=w, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index, MINIMIZE_GPU_MEMORY_USAGE=True, **kw)
    basis = BO.make_basis(N0, N1, ker[0], ker[1], ker[2], ker[3], bkg[0], bkg[1], bkg[2], bkg[3])
    scab = SV.make_scaling_basis(N0, N1, len(basis["ker_pairs"]), sca[0], sca[1], sca[2], sca[3])
    p = SV.SSC(N0, N1, w, basis, scab, "SEPARATE-VARYING")
    return pair, cfg, basis, scab, p


@pytest.mark.parametrize("c", SV_CASES, ids=[c["name"] for c in SV_CASES])
def test_varying_scaling_matches_oracle(dev, c):
    from oracle import bspline_sv_oracle as SV
    from sfft_amd.BSplineSFFT import GeneralSFFTSubtract as BGSS, ElementalSFFTSubtract as BESS
    pair, cfg, basis, scab, p = _sv_setup(c, dev)
    P = cfg[0]
    assert (P["NEQ"], P["NEQt"], P["ScaFij"], P["SCALING_MODE"]) == (p["NEQ"], p["NEQt"], p["ScaFij"], "SEPARATE-VARYING")
    plan = cfg[1]["plan"]
    assert plan.query("ScaFij") == p["ScaFij"]
    plan.solve(_to(dev, pair["mREF"]), _to(dev, pair["mSCI"]))
    LH, rhs = plan.get_system()
    LH_o, rhs_o = SV.establish_system(pair["mREF"], pair["mSCI"], p, basis, scab, workers=8)
    assert np.max(np.abs(LH.cpu().numpy() - LH_o)) <= 1e-11 * np.max(np.abs(LH_o))
    assert np.max(np.abs(rhs.cpu().numpy() - rhs_o)) <= 1e-11 * np.max(np.abs(rhs_o))
    # apply-only with the oracle's solution, then end to end
    sol_o = SV.solve_system(LH_o, rhs_o, p)
    D_o = SV.subtract(pair["REF"], pair["SCI"], sol_o, p, basis, scab, workers=8)
    D = BESS.ESS(pair["REF"], pair["SCI"], cfg, SFFTSolution=sol_o, Subtract=True, VERBOSE_LEVEL=0)[1]
    assert rms(D - D_o) <= 1e-10 * rms(pair["SCI"])
    sol, D2, _ = BGSS.GSS(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], cfg, VERBOSE_LEVEL=0)
    assert rel_rms_err(D2, D_o) <= 1e-6
    L = 2 * c["w"] + 1
    ij00 = np.arange(c["w"] * L + c["w"], P["Fijab"], L * L)
    assert np.all(sol[ij00[p["ScaFij"]:]] == 0.0)          # place-holder scaling terms come back as exact zeros
    assert plan.query("LAST_SOLVER") == 1


@pytest.mark.parametrize("mode", ["ENTANGLED", "SEPARATE-CONSTANT", "SEPARATE-VARYING"])
def test_kernel_regularisation_matches_oracle(dev, mode):
    from oracle import bspline_oracle as BO, bspline_sv_oracle as SV
    from sfft_amd.BSplineSFFT import SingleSFFTConfigure as BSSC, GeneralSFFTSubtract as BGSS
    from sfft_amd.utils.synthetic import make_pair
    N0, N1, w = 96, 80, 3
    pair = make_pair(N0, N1, seed=8, mask=True, density=400.0)
    ker = ("B-Spline", 2, [48.5], [])
    XY = np.array([[x, y] for x in (12.0, 48.0, 84.0) for y in (10.0, 40.0, 70.0)])
    W = np.linspace(1.0, 3.0, XY.shape[0])
    LAM = 200.0     # SCALE^2 makes REGMAT tiny: this lambda puts the penalty at the size of the kernel block of LHMAT
    kw = dict(KerSpType=ker[0], KerSpDegree=ker[1], KerIntKnotX=ker[2], KerIntKnotY=ker[3], BkgSpType="Polynomial", BkgSpDegree=1,
              REGULARIZE_KERNEL=True, IGNORE_LAPLACIAN_KERCENT=True, XY_REGULARIZE=XY, WEIGHT_REGULARIZE=W, LAMBDA_REGULARIZE=LAM)
    if mode == "ENTANGLED":
        kw.update(SEPARATE_SCALING=False)
    elif mode == "SEPARATE-CONSTANT":
        kw.update(SEPARATE_SCALING=True, ScaSpDegree=0)
    else:
        kw.update(SEPARATE_SCALING=True, ScaSpType="Polynomial", ScaSpDegree=1, MINIMIZE_GPU_MEMORY_USAGE=True)
    cfg = BSSC.SSC(NX=N0, NY=N1, KerHW=w, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index, **kw)
    plan = cfg[1]["plan"]
    basis = BO.make_basis(N0, N1, ker[0], ker[1], ker[2], ker[3], "Polynomial", 1, [], [])
    Fij = len(basis["ker_pairs"])
    scab = SV.make_scaling_basis(N0, N1, Fij, "Polynomial", 1) if mode == "SEPARATE-VARYING" else None
    p = SV.SSC(N0, N1, w, basis, scab, mode)
    kerspec = dict(KerSpType=ker[0], DK=ker[1], KerIntKnotX=ker[2], KerIntKnotY=ker[3])
    SST, CSST, DSST = SV.spatial_gram(p, kerspec, scab, XY, W)
    REG = SV.regularization_matrix(p, SV.laplacian_ireg(w, w, True), SST, CSST, DSST)
    if mode == "SEPARATE-VARYING":
        LH_o, rhs_o = SV.establish_system(pair["mREF"], pair["mSCI"], p, basis, scab, workers=8)
    else:
        LH_o, rhs_o = BO.establish_system(pair["mREF"], pair["mSCI"], BO.SSC(N0, N1, w, basis, mode != "ENTANGLED"), basis, workers=8)
    LH_o = LH_o + LAM * REG
    sol, D, _ = BGSS.GSS(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], cfg, VERBOSE_LEVEL=0)
    LH, rhs = plan.get_system()
    assert np.max(np.abs(LH.cpu().numpy() - LH_o)) <= 1e-11 * np.max(np.abs(LH_o))
    assert np.max(np.abs(rhs.cpu().numpy() - rhs_o)) <= 1e-11 * np.max(np.abs(rhs_o))
    nk = p["Fijab"]             # the penalty lives in the kernel block, whose entries are far smaller than the background block's
    assert np.max(np.abs(LH.cpu().numpy()[:nk, :nk] - LH_o[:nk, :nk])) <= 1e-11 * np.max(np.abs(LH_o[:nk, :nk]))
    assert np.max(np.abs(LAM * REG)) > 1e-2 * np.max(np.abs(LH_o[:nk, :nk]))
    sol_o = SV.solve_system(LH_o, rhs_o, p)
    if mode == "SEPARATE-VARYING":
        D_o = SV.subtract(pair["REF"], pair["SCI"], sol_o, p, basis, scab, workers=8)
    else:
        D_o = BO.subtract(pair["REF"], pair["SCI"], sol_o, BO.SSC(N0, N1, w, basis, mode != "ENTANGLED"), basis, workers=8)
    assert rel_rms_err(D, D_o) <= 1e-6
    # the penalty belongs to the config, not to the (shared, cached) plan: a second config of the same geometry without it
    # solves the plain system, and the first config still solves the penalised one afterwards (the reference derives REGMAT
    # from each config's own parameter dictionary at ESS time)
    from sfft_amd.BSplineSFFT import ElementalSFFTSubtract as BESS
    kw.update(REGULARIZE_KERNEL=False)
    cfg2 = BSSC.SSC(NX=N0, NY=N1, KerHW=w, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index, **kw)
    assert cfg2[1]["plan"] is plan
    sol2 = BESS.ESS(pair["mREF"], pair["mSCI"], cfg2, VERBOSE_LEVEL=0)[0]
    LH2, _ = plan.get_system()
    assert np.max(np.abs(LH2.cpu().numpy()[:nk, :nk] - (LH_o - LAM * REG)[:nk, :nk])) <= 1e-11 * np.max(np.abs(LH_o[:nk, :nk]))
    sol1 = BESS.ESS(pair["mREF"], pair["mSCI"], cfg, VERBOSE_LEVEL=0)[0]
    LH1, _ = plan.get_system()
    assert np.max(np.abs(LH1.cpu().numpy()[:nk, :nk] - LH_o[:nk, :nk])) <= 1e-11 * np.max(np.abs(LH_o[:nk, :nk]))
    assert np.array_equal(sol1, sol.cpu().numpy() if hasattr(sol, "cpu") else sol) and not np.array_equal(sol1, sol2)


def test_varying_scaling_large_shape_properties(dev):
    """2048 x 2048, KerHW 6, B-spline kernel (Fij = 9) with polynomial scaling of degree 1: too big for the oracle;
    the solution must satisfy its own (selected) linear system and DIFF must be the model residual of a second apply."""
    from sfft_amd.BSplineSFFT import SingleSFFTConfigure as BSSC
    from sfft_amd.utils.synthetic import make_pair
    N0 = N1 = 2048
    w = 6
    pair = make_pair(N0, N1, seed=12, mask=False)
    cfg = BSSC.SSC(NX=N0, NY=N1, KerHW=w, KerSpType="B-Spline", KerSpDegree=2, SEPARATE_SCALING=True, ScaSpType="Polynomial",
                   ScaSpDegree=1, BkgSpType="Polynomial", BkgSpDegree=2, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index)
    plan = cfg[1]["plan"]
    I, J = _to(dev, pair["REF"]), _to(dev, pair["SCI"])
    sol, diff = plan.subtract(I, J, I, J)
    LH, rhs = plan.get_system()
    P = cfg[0]
    L = 2 * w + 1
    ij00 = np.arange(w * L + w, P["Fijab"], L * L)
    keep = np.setdiff1d(np.arange(P["NEQ"]), ij00[P["ScaFij"]:])
    it = torch.from_numpy(keep).to(dev)
    r = LH[it][:, it] @ sol[it] - rhs[it]
    assert float(r.abs().max()) <= 1e-6 * float(rhs.abs().max())
    assert float(sol[torch.from_numpy(ij00[P["ScaFij"]:]).to(dev)].abs().max()) == 0.0
    d = diff.cpu().numpy()
    assert np.isfinite(d).all() and rms(d) < 2.0 * rms(pair["SCI"] - pair["REF"])
    # scaling at the image centre ~ the synthetic photometric ratio
    s = sol.cpu().numpy()[ij00[:3]] / (float(N0) * float(N1))
    assert abs(s[0] + 0.5 * s[1] + 0.5 * s[2] - 1.3) < 0.06      # REF_ij of degree 1: (0,0), (0,1), (1,0)
    del cfg, plan


# ------------------------------------------------------------------------------------------------
# (f) FFT utilities / noise decorrelation
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(64, 64), (276, 300), (4096, 128), (100, 4100), (6144, 48), (72, 9216), (40, 6144), (24, 4096), (16, 9232)])
def test_rfft2_irfft2_roundtrip_and_numpy(dev, shape):
    from sfft_amd.fftkit import get_fft_plan
    rng = np.random.default_rng(shape[0])
    x = rng.normal(size=shape)
    plan = get_fft_plan(shape[0], shape[1], dev.index)
    F = plan.rfft2(_to(dev, x))
    ref = np.fft.rfft2(x)
    assert np.max(np.abs(F.cpu().numpy() - ref)) <= 1e-12 * np.max(np.abs(ref))
    back = plan.irfft2(F).cpu().numpy()
    assert np.max(np.abs(back - x)) <= 1e-12 * np.max(np.abs(x))
    # the caller's factor rides on the spectrum copy (no pass of its own): unnormalised and an arbitrary one
    back1 = plan.irfft2(F, scale=1.0).cpu().numpy()
    assert np.max(np.abs(back1 - x * (shape[0] * shape[1]))) <= 1e-12 * np.max(np.abs(x)) * shape[0] * shape[1]
    back2 = plan.irfft2(F, scale=-2.5).cpu().numpy()
    assert np.max(np.abs(back2 + 2.5 * x * (shape[0] * shape[1]))) <= 1e-12 * np.max(np.abs(x)) * 2.5 * shape[0] * shape[1]


def test_decorrelation_kernel_matches_reference(dev):
    """DCC on the reference's own decorrelation test inputs against the reference's DCC output (and, transitively, its
    4check/DeCorrKernel.fits: see tests/golden/make_golden_decorr.py)."""
    import os
    from sfft_amd.utils.DeCorrelationCalculator import DeCorrelation_Calculator
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decorr_case.npz"))
    mkS = [None] + [z["mkS%d" % k] for k in range(1, 5)]
    mkR = [None] + [z["mkR%d" % k] for k in range(1, 5)]
    sigS = [float(z["sigS%d" % k]) for k in range(5)]
    sigR = [float(z["sigR%d" % k]) for k in range(5)]
    K = DeCorrelation_Calculator.DCC(MK_JLst=mkS, SkySig_JLst=sigS, MK_ILst=mkR, SkySig_ILst=sigR, MK_Fin=z["mkFin"], KERatio=2.0,
                                     VERBOSE_LEVEL=0, CUDA_DEVICE=dev.index)
    assert K.shape == z["KDeCo_sub"].shape
    assert np.max(np.abs(K - z["KDeCo_sub"])) <= 1e-11 * np.max(np.abs(z["KDeCo_sub"]))
    Ks = DeCorrelation_Calculator.DCC(MK_JLst=mkS, SkySig_JLst=sigS, KERatio=1.5, VERBOSE_LEVEL=0, CUDA_DEVICE=dev.index)
    assert np.max(np.abs(Ks - z["KDeCo_stack"])) <= 1e-11 * np.max(np.abs(z["KDeCo_stack"]))
    with pytest.raises(Exception, match="at least 2 J-images"):
        DeCorrelation_Calculator.DCC(MK_JLst=[z["mkFin"]], SkySig_JLst=[1.0], VERBOSE_LEVEL=0)


def test_pcdc_and_fft_convolve_match_oracle(dev):
    from oracle import decorr_oracle as DO
    from sfft_amd.utils.PureCupyDeCorrelationCalculator import PureCupy_DeCorrelation_Calculator as P
    from sfft_amd.utils.PureCupyFFTKits import PureCupy_FFTKits as FK
    rng = np.random.default_rng(8)
    g = lambda L: np.exp(-0.5 * (np.arange(L) - L // 2)[:, None] ** 2 / 2.0 - 0.5 * (np.arange(L) - L // 2)[None, :] ** 2 / 3.0) * (1 + 0.05 * rng.normal(size=(L, L)))
    KJ, KI, MK = [g(9), None, g(11)], [g(7), g(9)], g(13)
    sJ, sI = [2.0, 3.0, 2.5], [1.5, 1.0]
    NX, NY = 200, 144
    t = lambda a: None if a is None else _to(dev, a)
    F = P.PCDC(NX, NY, [t(k) for k in KJ], sJ, [t(k) for k in KI], sI, MATCH_KERNEL_GPU=t(MK), REAL_OUTPUT=False).cpu().numpy()
    Fo = DO.pcdc_fourier(NX, NY, KJ, sJ, KI, sI, MK)
    assert F.shape == (NX, NY) and np.max(np.abs(F - Fo)) <= 1e-12 * np.max(np.abs(Fo))
    img = rng.normal(size=(150, 131)) * 10
    img[5, 7] = np.nan
    K11 = np.exp(-0.5 * (np.arange(11) - 5)[:, None] ** 2 / 2.0 - 0.5 * (np.arange(11) - 5)[None, :] ** 2 / 3.0)
    out = FK.FFT_CONVOLVE(t(img), t(K11), PAD_FILL_VALUE=1.0, NAN_FILL_VALUE=0.0, NORMALIZE_KERNEL=True).cpu().numpy()
    ref = DO.fft_convolve(img.copy(), K11, 1.0, 0.0, True)
    assert out.shape == img.shape and np.max(np.abs(out - ref)) <= 1e-11 * np.max(np.abs(ref))


# ------------------------------------------------------------------------------------------------
# (g) B-spline post-processing on the GPU: decorrelation kernel from realised kernels, grid convolution
# ------------------------------------------------------------------------------------------------
def test_bspline_decorrelation_matches_reference(dev):
    """BSpline_DeCorrelation.BDC against the reference's own output (tests/golden/make_golden_bspline_post.py)."""
    from sfft_amd.BSplineSFFT import BSpline_DeCorrelation
    from _golden import GOLDEN_DIR
    G = np.load(os.path.join(GOLDEN_DIR, "bspline_post_cases.npz"), allow_pickle=False)
    mkj, mki, mkf = G["bdc_mkj"], G["bdc_mki"], G["bdc_mkf"]
    out = BSpline_DeCorrelation.BDC(MK_JLst=[mkj], SkySig_JLst=[3.0], MK_ILst=[mki], SkySig_ILst=[2.0], MK_Fin=mkf, KERatio=2.0,
                                    DENO_CLIP_RATIO=100000.0, VERBOSE_LEVEL=0, CUDA_DEVICE=dev.index)
    assert out.shape == G["bdc_sub"].shape and np.abs(out - G["bdc_sub"]).max() <= 1e-11 * np.abs(G["bdc_sub"]).max()
    out = BSpline_DeCorrelation.BDC(MK_JLst=[None], SkySig_JLst=[3.0], MK_ILst=[mki], SkySig_ILst=[2.0], MK_Fin=None, KERatio=1.5,
                                    DENO_CLIP_RATIO=1000.0, VERBOSE_LEVEL=0, CUDA_DEVICE=dev.index)
    assert out.shape == G["bdc_sub_nofin"].shape and np.abs(out - G["bdc_sub_nofin"]).max() <= 1e-11 * np.abs(G["bdc_sub_nofin"]).max()
    out = BSpline_DeCorrelation.BDC(MK_JLst=[mkj, None, mkf], SkySig_JLst=[3.0, 2.5, 4.0], KERatio=2.0, VERBOSE_LEVEL=0, CUDA_DEVICE=dev.index)
    assert out.shape == G["bdc_stack"].shape and np.abs(out - G["bdc_stack"]).max() <= 1e-11 * np.abs(G["bdc_stack"]).max()
    with pytest.raises(Exception, match="at least 2 J-images"):
        BSpline_DeCorrelation.BDC(MK_JLst=[mkj], SkySig_JLst=[3.0], VERBOSE_LEVEL=0, CUDA_DEVICE=dev.index)


@pytest.mark.parametrize("shape,TiHW,L,norm", [((200, 170), 10, (7, 7), True), ((96, 131), 7, (5, 9), False), ((64, 64), 31, (21, 21), True)])
def test_grid_convolve_matches_oracle(dev, shape, TiHW, L, norm):
    """BSpline_GridConvolve.GSVC_GPU against the scipy restatement of the reference's loop (pinned only through the NIRCam chain: the reference
    needs CuPy / astropy); both of its branches (direct and FFT) are reproduced by the same direct sums."""
    from oracle import gridconv_oracle as GO
    from sfft_amd.BSplineSFFT import BSpline_GridConvolve
    rng = np.random.default_rng(shape[0])
    N0, N1 = shape
    img = rng.normal(size=shape) + 10.0
    img[3, 5] = np.nan
    lab, XY = GO.tile_labels(N0, N1, TiHW)
    Nseg = int(lab.max()) + 1
    kers = rng.uniform(0.1, 1.0, size=(Nseg, L[0], L[1]))           # asymmetric kernels: convolution, not correlation
    gc = BSpline_GridConvolve(img, lab, kers, nan_fill_value=0.0, use_fft=False, normalize_kernel=norm)
    out = gc.GSVC_GPU(CUDA_DEVICE=str(dev.index))
    ref = GO.gsvc(gc.PixA_in, lab, kers, normalize_kernel=norm, use_fft=False)
    assert np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max()
    ref_fft = GO.gsvc(gc.PixA_in, lab, kers, normalize_kernel=norm, use_fft=True)
    assert np.abs(out - ref_fft).max() <= 1e-10 * np.abs(ref).max()
    with pytest.raises(Exception, match="GPU variant"):
        gc.GSVC_CPU()


def test_matching_kernel_grid_convolve_reproduces_sfft_model(dev):
    """Ties the post-processing to the core: realise the solved kernel on a tile grid, grid-convolve the reference image with
    it, and compare with what the subtraction itself removed, J - DIFF - background (spatially constant kernel -> the tile
    kernels are all equal and the two agree to rounding away from the image border, where SFFT wraps and GSVC zero-fills)."""
    from oracle import gridconv_oracle as GO
    from sfft_amd.BSplineSFFT import SingleSFFTConfigure as BSSC, GeneralSFFTSubtract as BGSS, BSpline_MatchingKernel, BSpline_GridConvolve
    from sfft_amd.utils.synthetic import make_pair
    N0, N1, w = 160, 128, 3
    pair = make_pair(N0, N1, seed=19, mask=False, density=300.0)
    kw = dict(KerSpType="Polynomial", KerSpDegree=0, SEPARATE_SCALING=False, BkgSpType="Polynomial", BkgSpDegree=1)
    cfg = BSSC.SSC(NX=N0, NY=N1, KerHW=w, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index, **kw)
    sol, D, _ = BGSS.GSS(pair["REF"], pair["SCI"], pair["REF"], pair["SCI"], cfg, VERBOSE_LEVEL=0)
    P = cfg[0]
    lab, XY = GO.tile_labels(N0, N1, 12)
    ks = BSpline_MatchingKernel(XY, VERBOSE_LEVEL=0).FromArray(
        Solution=sol, KerSpType="Polynomial", KerIntKnotX=[], KerIntKnotY=[], N0=N0, N1=N1, DK=0, L0=P["L0"], L1=P["L1"], Fi=-1, Fj=-1,
        Fpq=P["Fpq"], SEPARATE_SCALING=False, ScaSpType=None, ScaIntKnotX=None, ScaIntKnotY=None, DS=None, ScaFi=None, ScaFj=None)
    conv = BSpline_GridConvolve(pair["REF"], lab, ks, normalize_kernel=False).GSVC_GPU(CUDA_DEVICE=str(dev.index))
    cx = (np.arange(N0)[:, None] + 1.0) / N0
    cy = (np.arange(N1)[None, :] + 1.0) / N1
    b = sol[P["Fijab"]:]
    bkg = b[0] + b[1] * cy + b[2] * cx                  # REF_pq of degree 1: (0,0), (0,1), (1,0)
    model = pair["SCI"] - D - bkg
    inner = (slice(w, N0 - w), slice(w, N1 - w))
    assert np.abs(conv[inner] - model[inner]).max() <= 1e-9 * np.abs(model).max()


# ------------------------------------------------------------------------------------------------
# (h) alternative formulations on the 4096^2 fast path must agree with each other
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,DK,DB,cpr", [(8, 2, 2, True), (3, 3, 1, False), (5, 1, 2, True), (8, 0, 0, True)])
def test_mixed_domain_apply_equals_fourier_apply_4096(dev, w, DK, DB, cpr):
    """The apply pass of polynomial plans on the staged fast path convolves the stage planes along the columns in the mixed
    (row index, column frequency) domain (vconv_mixed) instead of transforming 6 planes and multiplying in Fourier space.
    Both are exact restatements of Construct_FDIFF + inverse DFT: same DIFF to rounding for a random solution, and the whole
    GSS gives the same answer either way."""
    from sfft_amd.plan import Plan
    from sfft_amd.utils.synthetic import make_pair
    N = 4096
    pair = make_pair(N, N, seed=77 + w, mask=True)
    I, J = _to(dev, pair["REF"]), _to(dev, pair["SCI"])
    mI, mJ = _to(dev, pair["mREF"]), _to(dev, pair["mSCI"])
    outs = []
    for no_vconv in ("", "1"):
        if no_vconv:
            os.environ["SFFT_NO_VCONV"] = "1"
        else:
            os.environ.pop("SFFT_NO_VCONV", None)
        os.environ["SFFT_COLZ"] = "0"       # (the pair-per-workgroup column pass belongs to mixed-domain plans only: here both plans take the four-column kernel)
        try:
            plan = Plan(N, N, w, DK, DB, cpr, device=dev.index)
        finally:
            os.environ.pop("SFFT_NO_VCONV", None)
            os.environ.pop("SFFT_COLZ", None)
        rng = np.random.default_rng(5)
        sol = rng.normal(size=plan.NEQ)
        sol[:plan.Fijab] *= float(N) * float(N) * 0.01
        d_rand = plan.apply(I, J, torch.from_numpy(sol).to(dev)).cpu().numpy()
        s2, d2 = plan.subtract(I, J, mI, mJ)
        outs.append((d_rand, s2.cpu().numpy(), d2.cpu().numpy()))
        plan.close()
    (da, sa, ga), (db, sb, gb) = outs
    assert rms(da - db) <= 1e-12 * rms(db)
    assert np.array_equal(sa, sb)                       # the solve pass is the same code either way
    assert rms(ga - gb) <= 1e-11 * rms(pair["SCI"])


@pytest.mark.parametrize("shape,w,DK", [((96, 66), 4, 2), ((80, 62), 3, 1), ((72, 64), 8, 2), ((130, 98), 8, 3), ((200, 34), 2, 0),
                                        ((4096, 4096), 8, 2), ((1536, 2048), 6, 2)])
def test_mixed_domain_apply_variants_agree(dev, shape, w, DK):
    """The mixed-domain apply kernels against each other and against the Fourier-domain apply, on shapes whose last 16-column
    tile holds 2 / 0 / 1 / 2 / 2 / 1 / 1 columns (1 or 2: taken point by point, inside the main launch or by vconv_direct): the default (vconv_mixed2 with the stream
    length balanced against the CU count), the round-1 stream length, one source row per table read (vconv_mixed), the
    Fourier-domain apply (construct_fd)."""
    from sfft_amd.plan import Plan
    from sfft_amd.utils.synthetic import make_pair
    pair = make_pair(*shape, seed=11 + w, mask=True)
    I, J = _to(dev, pair["REF"]), _to(dev, pair["SCI"])
    rng = np.random.default_rng(9)
    outs = {}
    for name, env in [("default", {}), ("r1_len", {"SFFT_VCONV_R": "0"}), ("one_row", {"SFFT_VCONV_RP": "1"}),
                      ("short", {"SFFT_VCONV_R": "17"}), ("own_launch", {"SFFT_VCONV_DIRECT": "1"}), ("fourier", {"SFFT_NO_VCONV": "1"})]:
        for k, v in env.items():
            os.environ[k] = v
        try:
            plan = Plan(shape[0], shape[1], w, DK, 1, True, device=dev.index)
        finally:
            for k in env:
                os.environ.pop(k, None)
        if name == "default":
            sol = rng.normal(size=plan.NEQ)
            sol[:plan.Fijab] *= float(shape[0]) * float(shape[1]) * 0.01
        outs[name] = plan.apply(I, J, torch.from_numpy(sol).to(dev)).cpu().numpy()
        plan.close()
    ref = outs["fourier"]
    for name, d in outs.items():
        assert rms(d - ref) <= 1e-12 * rms(ref), name
    for name in ("r1_len", "short", "one_row", "own_launch"):          # the same tap sums (vconv_direct adds the taps in the opposite order: rounding only)
        assert rms(outs[name] - outs["default"]) <= 1e-14 * rms(ref), name


def _subtract_with_env(dev, env, shape, w, DK, DB, pair):
    """One GSS with a plan created under the given environment switches (they are read at plan creation)."""
    from sfft_amd.plan import Plan
    for k, v in env.items():
        os.environ[k] = v
    try:
        plan = Plan(shape[0], shape[1], w, DK, DB, True, device=dev.index)
    finally:
        for k in env:
            os.environ.pop(k, None)
    I, J = _to(dev, pair["REF"]), _to(dev, pair["SCI"])
    mI, mJ = _to(dev, pair["mREF"]), _to(dev, pair["mSCI"])
    sol, diff = plan.subtract(I, J, mI, mJ)
    plan.solve(mI, mJ)
    LH, rhs = plan.get_system()
    out = (sol.cpu().numpy(), diff.cpu().numpy(), LH.cpu().numpy(), rhs.cpu().numpy())
    plan.close()
    return out


@pytest.mark.parametrize("w,DK,DB,same", [(8, 2, 2, False), (4, 1, 0, False), (8, 2, 2, True), (12, 3, 2, False), (5, 2, 1, False), (2, 0, 0, False)])
def test_pair_column_pass_equals_quad_column_pass_4096(dev, w, DK, DB, same):
    """Round 6: the solve pass of the 4096^2 path transforms one column PAIR per workgroup (cols_fwd_weighted_4096_z: two workgroups per CU,
    stage planes with pair-major lines, spectra in 2-column panels read by the Greek launches) -- against the four-column kernel
    (SFFT_COLZ=0, the round-1..5 path, itself held to the oracle by test_config2_full_size_matches_oracle): the same linear system block by
    block to 1e-12, the same difference image.  KerHW 12: the Omega launch in two lag bands; KerHW 5 / 2: the vector Greek kernels -- every reader of the
    2-column spectra.  `same`: I passed as its own mask (the apply pass then must NOT reuse the solve pass's
    pair-major stage planes).  A self-comparison of two HIP paths: a regression guard, not parity evidence."""
    from sfft_amd.utils.synthetic import make_pair
    shape = (4096, 4096)
    pair = make_pair(*shape, seed=77 + w, mask=True)
    if same:
        pair = dict(pair); pair["mREF"] = pair["REF"]
    def run(env):
        from sfft_amd.plan import Plan
        for k, v in env.items():
            os.environ[k] = v
        try:
            plan = Plan(shape[0], shape[1], w, DK, DB, True, device=dev.index)
        finally:
            for k in env:
                os.environ.pop(k, None)
        I, J, mJ = _to(dev, pair["REF"]), _to(dev, pair["SCI"]), _to(dev, pair["mSCI"])
        mI = I if same else _to(dev, pair["mREF"])
        plan.set_timing(True)
        sol, diff = plan.subtract(I, J, mI, mJ)
        LH, rhs = plan.get_system()
        kernels = plan.stage_kernels()["fwd_cols"]
        out = (sol.cpu().numpy(), diff.cpu().numpy(), LH.cpu().numpy(), rhs.cpu().numpy(), kernels)
        plan.close()
        return out
    a = run({})
    b = run({"SFFT_COLZ": "0"})
    assert a[4] == ["cols_fwd_weighted_4096_z"] and b[4] == ["cols_fwd_weighted_4096_q"], (a[4], b[4])
    nk = (DK + 1) * (DK + 2) // 2 * (2 * w + 1) ** 2
    for blk in (np.s_[:nk, :nk], np.s_[:nk, nk:], np.s_[nk:, nk:]):
        assert np.abs(a[2][blk] - b[2][blk]).max() <= 1e-12 * np.abs(b[2][blk]).max()
    for blk in (np.s_[:nk], np.s_[nk:]):
        assert np.abs(a[3][blk] - b[3][blk]).max() <= 1e-12 * np.abs(b[3][blk]).max()
    assert rms(a[1] - b[1]) <= 1e-8 * rms(b[1])


def test_pair_column_pass_with_a_bspline_basis_4096(dev):
    """The pair-per-workgroup column pass under a tabulated (B-spline, 5 x 5 terms) kernel basis at 4096^2: 26 output planes from 6 stage planes in
    one launch, Omega products summed in real space beside it (omega_sparse), the tensor form of the mixed-domain apply -- against the four-column
    kernel (SFFT_COLZ=0), itself held to the oracle on the 768^2 / 1536^2 / 6144 x 96 B-spline cases.  A self-comparison: regression guard."""
    from sfft_amd.plan import Plan
    from sfft_amd.BSplineSFFT import _axis_tables
    N, w = 4096, 4
    REF, SCI, mREF, mSCI = _blob_pair(N, N, 9)
    kx, ky = [N / 3 + 0.5, 2 * N / 3 + 0.5], [N / 3 + 0.5, 2 * N / 3 + 0.5]
    kbx, kby, kpairs = _axis_tables(N, N, "B-Spline", 2, kx, ky)
    tbx, tby, bpairs = _axis_tables(N, N, "Polynomial", 2, [], [])
    bdict = dict(kbx=kbx, kby=kby, ker_pairs=kpairs, tbx=tbx, tby=tby, bkg_pairs=bpairs, scaling_mode=2)
    I, J, mI, mJ = _to(dev, REF), _to(dev, SCI), _to(dev, mREF), _to(dev, mSCI)
    outs = []
    for env in ({}, {"SFFT_COLZ": "0"}):
        for k, v in env.items():
            os.environ[k] = v
        try:
            plan = Plan(N, N, w, device=dev.index, basis=bdict)
        finally:
            for k in env:
                os.environ.pop(k, None)
        plan.set_timing(True)
        sol, diff = plan.subtract(I, J, mI, mJ)
        LH, rhs = plan.get_system()
        outs.append((sol.cpu().numpy(), diff.cpu().numpy(), LH.cpu().numpy(), rhs.cpu().numpy(), plan.stage_kernels()["fwd_cols"]))
        plan.close()
    a, b = outs
    assert a[4] == ["cols_fwd_weighted_4096_z"] and b[4] == ["cols_fwd_weighted_4096_q"], (a[4], b[4])
    assert np.abs(a[2] - b[2]).max() <= 1e-12 * np.abs(b[2]).max() and np.abs(a[3] - b[3]).max() <= 1e-12 * np.abs(b[3]).max()
    assert rms(a[1] - b[1]) <= 1e-8 * rms(b[1])


def test_outer_blocked_cholesky_equals_plain_blocked(dev):
    """Systems of 3000+ unknowns factor in 256-column outer blocks with a rank-256 matrix-core update (chol_syrk); forced on
    at n = 1735 it must give the solution of the rank-64 path on the same matrix (and the matrix itself is untouched)."""
    from sfft_amd.utils.synthetic import make_pair
    shape = (320, 288)
    pair = make_pair(*shape, seed=31, mask=True)
    a = _subtract_with_env(dev, {"SFFT_CHOL_OUTER_MIN": "100000"}, shape, 8, 2, 2, pair)
    b = _subtract_with_env(dev, {"SFFT_CHOL_OUTER_MIN": "256"}, shape, 8, 2, 2, pair)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    # both are backward-stable factorisations of the same SPD matrix: compare through the residual, not digit by digit
    LH, rhs = a[2], a[3]
    for sol in (a[0], b[0]):
        keep = np.flatnonzero(sol != 0.0)            # (the stripes removed for the constant photometric ratio stay exactly 0)
        x = sol[keep]
        r = LH[np.ix_(keep, keep)] @ x - rhs[keep]
        assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(rhs[keep])
    assert rms(a[1] - b[1]) <= 1e-7 * rms(a[1])


@pytest.mark.parametrize("shape,w", [((1536, 768), 4), ((300, 500), 3), ((2048, 1024), 5), ((96, 1000), 2)])
def test_generic_fft_variants_agree(dev, shape, w):
    """Generic shapes: radix-16 / mixed-radix on-chip stages, folded Bluestein products, panel planes and staged forward
    transforms against the first formulation of each (radix-4 stages, separate pointwise passes, row-major planes, one row
    transform per plane).  Same linear system to rounding, same DIFF."""
    from sfft_amd.utils.synthetic import make_pair
    pair = make_pair(*shape, seed=shape[0] + w, mask=True)
    new = _subtract_with_env(dev, {}, shape, w, 2, 1, pair)
    old = _subtract_with_env(dev, {"SFFT_NO_R16": "1", "SFFT_NO_MIXED_RADIX": "1", "SFFT_PANEL": "0", "SFFT_NO_STAGED": "1"},
                             shape, w, 2, 1, pair)
    assert np.max(np.abs(new[2] - old[2])) <= 1e-11 * np.max(np.abs(old[2]))
    assert np.max(np.abs(new[3] - old[3])) <= 1e-11 * np.max(np.abs(old[3]))
    assert rms(new[1] - old[1]) <= 1e-7 * rms(old[1])


@pytest.mark.parametrize("env", [{"SFFT_NO_RADER_R24": "1"}, {"SFFT_NO_RADER": "1"}, {"SFFT_NO_DFT16_REGS": "1"}, {"SFFT_NO_STAGED": "1"}, {"SFFT_NO_VCONV": "1"},
                                 {"SFFT_NO_VCONV": "1", "SFFT_NO_RADER": "1", "SFFT_NO_DFT16_REGS": "1"}],
                         ids=["rader577_lds", "bluestein577", "lds16", "unstaged", "fourier_apply", "fourier_apply_round2_kernels"])
def test_four_step_column_axis_variants_agree(dev, env):
    """9232 = 16 x 577 rows (config 5's column axis): Rader's 576-point sub-transform against Bluestein on 2048 points, the
    register-only 16-point first pass against the LDS one, staged against per-plane forward transforms, and the Fourier-domain
    apply -- whose INVERSE column transform runs the same two kernels with the conjugation flags -- against the mixed-domain one.
    Same linear system to rounding, same DIFF."""
    from sfft_amd.utils.synthetic import make_pair
    shape, w = (9232, 80), 3
    pair = make_pair(*shape, seed=77, mask=True, density=400.0)
    new = _subtract_with_env(dev, {}, shape, w, 2, 1, pair)
    old = _subtract_with_env(dev, env, shape, w, 2, 1, pair)
    assert np.max(np.abs(new[2] - old[2])) <= 1e-11 * np.max(np.abs(old[2]))
    assert np.max(np.abs(new[3] - old[3])) <= 1e-11 * np.max(np.abs(old[3]))
    assert rms(new[1] - old[1]) <= 1e-7 * rms(old[1])


@pytest.mark.parametrize("shape", [(96, 6144), (80, 9216)])
def test_r24_inverse_row_pass_agrees_with_generic_pass(dev, shape):
    """SFFT_INV_R24 (default: on for 6144-point rows, off for 9216): the register-resident inverse row pass of 6144- / 9216-point rows
    (rows_c2r_diff_r24) gives the generic pass's difference image."""
    from sfft_amd.utils.synthetic import make_pair
    pair = make_pair(*shape, seed=62, mask=True, density=400.0)
    ref = _subtract_with_env(dev, {"SFFT_INV_R24": "0"}, shape, 3, 2, 1, pair)
    alt = _subtract_with_env(dev, {"SFFT_INV_R24": "1"}, shape, 3, 2, 1, pair)
    assert rms(alt[1] - ref[1]) <= 1e-12 * max(rms(ref[1]), 1.0) + 1e-10 * rms(pair["SCI"])
    assert np.array_equal(alt[0], ref[0])


@pytest.mark.parametrize("shape", [(6144, 96), (96, 6144), (9216, 80), (80, 9216)])
def test_r24_axis_kernels_agree_with_generic_passes(dev, shape):
    """6144- and 9216-point axes (configs 3 and 5): the register-resident 16 x 16 x 24 / 16 x 24 x 24 column and row kernels
    (fft_r24.hpp) against the generic LDS-resident passes they replace.  Same linear system to rounding, same DIFF."""
    from sfft_amd.utils.synthetic import make_pair
    w = 3
    pair = make_pair(*shape, seed=61, mask=True, density=400.0)
    new = _subtract_with_env(dev, {}, shape, w, 2, 1, pair)
    old = _subtract_with_env(dev, {"SFFT_NO_COLS_R24": "1", "SFFT_NO_ROWS_R24": "1"}, shape, w, 2, 1, pair)
    assert np.max(np.abs(new[2] - old[2])) <= 1e-11 * np.max(np.abs(old[2]))
    assert np.max(np.abs(new[3] - old[3])) <= 1e-11 * np.max(np.abs(old[3]))
    assert rms(new[1] - old[1]) <= 1e-7 * rms(old[1])
    if shape[0] == 6144:     # the 6144-point column kernel runs two columns per workgroup (neighbouring lanes); one column per workgroup is the same arithmetic
        one = _subtract_with_env(dev, {"SFFT_COLS_R24_PAIR": "0"}, shape, w, 2, 1, pair)
        assert np.array_equal(new[2], one[2]) and np.array_equal(new[1], one[1])


@pytest.mark.parametrize("shape,w,DK", [((256, 288), 8, 2), ((320, 4096), 8, 2), ((512, 384), 5, 3), ((384, 96), 12, 3)])
def test_omega_launch_variants_agree(dev, shape, w, DK):
    """Regression guard for the two remaining switches of the Omega + Theta launch and the apply pass (self-comparisons, not parity
    evidence): SFFT_G1_S=2 (two row chunks), SFFT_THETA_SLOTS=0 (KerHW 9 .. 16: the Theta passes in a launch of their own instead of
    slots of the first Omega launch; the KerHW 12 case here) and SFFT_VCONV2_W12=0 (KerHW 9 .. 12: the one-row tap walk of the apply
    pass).  Same system, same difference image."""
    from sfft_amd.utils.synthetic import make_pair
    pair = make_pair(*shape, seed=5 + w, mask=True, density=400.0)
    ref = _subtract_with_env(dev, {}, shape, w, DK, 1, pair)
    for env in ({"SFFT_G1_S": "2"}, {"SFFT_THETA_SLOTS": "0"}, {"SFFT_VCONV2_W12": "0"}):
        alt = _subtract_with_env(dev, env, shape, w, DK, 1, pair)
        assert np.max(np.abs(alt[2] - ref[2])) <= 1e-11 * np.max(np.abs(ref[2])), env
        assert np.max(np.abs(alt[3] - ref[3])) <= 1e-11 * np.max(np.abs(ref[3])), env
        assert rms(alt[1] - ref[1]) <= 1e-7 * rms(ref[1]), env


def test_solver_chain_replays_as_graph_on_a_side_stream(dev):
    """On a capturable stream the ~35 launches of the factorisation and back substitution are captured once per plan and
    replayed with hipGraphLaunch (flag stamps come from a device counter, so the arguments are constant).  Same solution as
    the plain launches on the default stream, call after call."""
    from sfft_amd.plan import Plan
    from sfft_amd.utils.synthetic import make_pair
    shape = (256, 320)
    pair = make_pair(*shape, seed=8, mask=True)
    I, J = _to(dev, pair["REF"]), _to(dev, pair["SCI"])
    mI, mJ = _to(dev, pair["mREF"]), _to(dev, pair["mSCI"])
    plan = Plan(shape[0], shape[1], 4, 2, 2, True, device=dev.index)
    s0, d0 = plan.subtract(I, J, mI, mJ)                      # legacy default stream: plain launches
    assert plan.query("SOLVE_GRAPH") == 0
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    outs = []
    with torch.cuda.stream(side):
        for _ in range(3):
            s1, d1 = plan.subtract(I, J, mI, mJ)
            outs.append((s1.clone(), d1.clone()))
    side.synchronize()
    assert plan.query("SOLVE_GRAPH") == 1
    for s1, d1 in outs:
        assert np.array_equal(s1.cpu().numpy(), s0.cpu().numpy())
        assert np.array_equal(d1.cpu().numpy(), d0.cpu().numpy())
    plan.close()


def test_lu_redo_after_a_failed_cholesky_attempt_inside_subtract(dev):
    """sfft_subtract enqueues solve and apply with one host sync at the end; when the status word then says that the Cholesky
    attempt met a non-positive pivot, the system is redone with pivoted LU and the apply pass is run again.  The failure is
    injected (SFFT_TEST_FAIL_CHOL); the result must be the ordinary one."""
    from sfft_amd.utils.synthetic import make_pair
    shape = (200, 168)
    pair = make_pair(*shape, seed=12, mask=True)
    ok = _subtract_with_env(dev, {}, shape, 3, 2, 1, pair)
    from sfft_amd.plan import Plan
    os.environ["SFFT_TEST_FAIL_CHOL"] = "1"
    try:
        plan = Plan(shape[0], shape[1], 3, 2, 1, True, device=dev.index)
    finally:
        os.environ.pop("SFFT_TEST_FAIL_CHOL", None)
    I, J = _to(dev, pair["REF"]), _to(dev, pair["SCI"])
    mI, mJ = _to(dev, pair["mREF"]), _to(dev, pair["mSCI"])
    for masks in ((mI, mJ), (I, J)):                     # overlapped path, and the path that reuses the solve pass's spectra
        sol, diff = plan.subtract(I, J, masks[0], masks[1])
        assert plan.query("LAST_SOLVER") == 2
        if masks[0] is mI:
            assert np.linalg.norm(sol.cpu().numpy() - ok[0]) <= 1e-6 * np.linalg.norm(ok[0])
            assert rms(diff.cpu().numpy() - ok[1]) <= 1e-7 * rms(ok[1])
        else:
            assert np.isfinite(diff.cpu().numpy()).all()
    plan.close()


# ------------------------------------------------------------------------------------------------
# (h) host-side sharing rules (round-1 advisor findings): configs of one geometry share a cached plan
# ------------------------------------------------------------------------------------------------
def test_two_host_threads_sharing_one_cached_plan_get_single_thread_results(dev):
    """Two threads run GSS through configs that share ONE cached plan (same geometry): the per-plan lock serialises them and each
    gets exactly the result of a run on its own."""
    import threading
    from sfft_amd.sfftcore import SingleSFFTConfigure, GeneralSFFTSubtract
    from sfft_amd.utils.synthetic import make_pair
    N0, N1, w = 192, 160, 3
    pairs = [make_pair(N0, N1, seed=40 + k, mask=True, sky=0.0, bkg_scale=0.05) for k in range(2)]
    cfgs = [SingleSFFTConfigure.SSC(N0, N1, w, 2, 1, True, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index) for _ in range(2)]
    assert cfgs[0][1]["plan"] is cfgs[1][1]["plan"]
    ref = [GeneralSFFTSubtract.GSS(p["REF"], p["SCI"], p["mREF"], p["mSCI"], cfgs[0], VERBOSE_LEVEL=0) for p in pairs]
    out = [[None] * 6 for _ in range(2)]

    def work(k):
        torch.cuda.set_device(dev)
        for it in range(6):
            out[k][it] = GeneralSFFTSubtract.GSS(pairs[k]["REF"], pairs[k]["SCI"], pairs[k]["mREF"], pairs[k]["mSCI"], cfgs[k], VERBOSE_LEVEL=0)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(2):
        for it in range(6):
            assert np.array_equal(out[k][it][0], ref[k][0]) and np.array_equal(out[k][it][1], ref[k][1])


def test_clearing_the_plan_cache_keeps_live_configs_usable(dev):
    from sfft_amd.plan import clear_plan_cache
    from sfft_amd.sfftcore import SingleSFFTConfigure, ElementalSFFTSubtract
    from sfft_amd.utils.synthetic import make_pair
    pair = make_pair(64, 48, seed=2, mask=False)
    cfg = SingleSFFTConfigure.SSC(64, 48, 2, 1, 1, True, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index)
    s0 = ElementalSFFTSubtract.ESS(pair["mREF"], pair["mSCI"], cfg, VERBOSE_LEVEL=0)[0]
    clear_plan_cache()
    s1 = ElementalSFFTSubtract.ESS(pair["mREF"], pair["mSCI"], cfg, VERBOSE_LEVEL=0)[0]       # the config still holds its plan
    assert np.array_equal(s0, s1)
    cfg2 = SingleSFFTConfigure.SSC(64, 48, 2, 1, 1, True, VERBOSE_LEVEL=0, CUDA_DEVICE_4SUBTRACT=dev.index)
    assert cfg2[1]["plan"] is not cfg[1]["plan"]
    assert np.array_equal(ElementalSFFTSubtract.ESS(pair["mREF"], pair["mSCI"], cfg2, VERBOSE_LEVEL=0)[0], s0)


def test_entry_points_leave_the_current_device_alone(dev):
    """Every C entry point restores the calling thread's HIP device (one device here: the call must at least not disturb it, and
    tensors created afterwards land on the device torch considers current)."""
    from sfft_amd.plan import get_plan
    from sfft_amd.utils.synthetic import make_pair
    pair = make_pair(64, 64, seed=1, mask=False)
    before = torch.cuda.current_device()
    plan = get_plan(64, 64, 2, 1, 1, True, dev.index)
    plan.subtract(_to(dev, pair["REF"]), _to(dev, pair["SCI"]), _to(dev, pair["mREF"]), _to(dev, pair["mSCI"]))
    assert torch.cuda.current_device() == before and torch.zeros(1, device="cuda").device.index == before


def test_pccp_accepts_host_tensors_with_nans(dev):
    """Inputs on the host (or another device) are moved to CUDA_DEVICE_4SUBTRACT before the NaN mask is built (advisor finding)."""
    from sfft_amd import PureCupy_Customized_Packet
    g = load_golden("c96x80_w3_k2b2_cpr_nan")
    m = g["meta"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))          # CPU tensors
    sol, diff = PureCupy_Customized_Packet.PCCP(t(g["REF"]), t(g["SCI"]), t(g["mREF"]), t(g["mSCI"]), m["ForceConv"], m["KerHW"],
                                                KerPolyOrder=m["DK"], BGPolyOrder=m["DB"], ConstPhotRatio=bool(m["CPR"]),
                                                CUDA_DEVICE_4SUBTRACT=str(dev.index), VERBOSE_LEVEL=0)
    assert diff.is_cuda and rel_rms_err(diff.cpu().numpy(), g["DIFF"]) <= 1e-6


def test_removed_unknowns_stay_zero_under_concurrent_graph_replays(dev):
    """Regression guard for a runtime fault found in round 3: with the Solution zeroed by a memset NODE of the captured solver graph,
    several plans replaying their graphs from different host threads now and then left the removed unknowns ij00[1:] (which
    Extend_Solution leaves at exactly 0, sfft/sfftcore/SFFTConfigure.py:716-732) holding stale bytes.  They are zeroed by a kernel node
    now; this test runs four plans / streams / threads over NaN-prefilled outputs and asserts exact zeros every time."""
    import threading
    from sfft_amd.plan import Plan
    from sfft_amd.utils.synthetic import make_pair
    N0 = N1 = 1024
    w, S, REPS = 4, 4, 12
    pair = make_pair(N0, N1, seed=3, mask=True)
    g = {k: _to(dev, v) for k, v in pair.items()}
    plans = [Plan(N0, N1, w, 2, 2, True, device=dev.index) for _ in range(S)]
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    NEQ, Fab = plans[0].NEQ, (2 * w + 1) ** 2
    forb = [ij * Fab + w * (2 * w + 1) + w for ij in range(1, 6)]
    bad, sols = [0] * S, [None] * S

    def worker(wi):
        torch.cuda.set_device(dev.index)
        with torch.cuda.stream(streams[wi]):
            for _ in range(REPS):
                sol = torch.full((NEQ,), float("nan"), dtype=torch.float64, device=dev)
                diff = torch.empty((N0, N1), dtype=torch.float64, device=dev)
                plans[wi].subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"], out_solution=sol, out_diff=diff)
                v = sol.cpu().numpy()
                if not np.all(v[forb] == 0.0) or not np.isfinite(v).all():
                    bad[wi] += 1
                sols[wi] = v
    th = [threading.Thread(target=worker, args=(i,)) for i in range(S)]
    [t.start() for t in th]
    [t.join() for t in th]
    for pl in plans:
        pl.close()
    assert bad == [0] * S, bad
    assert all(np.array_equal(sols[0], v) for v in sols[1:])          # every plan, every thread: the same bits


def test_stage_timing_names_the_kernels_that_ran(dev):
    """sfft_set_timing / sfft_stage_ms / sfft_stage_kernels: with timing on, every stage of a GSS reports a duration and the HIP kernels it
    launched, as written at the launch sites -- the names bench.py puts into its roofline objects and scripts/make_pmc_traffic.py uses to
    map PMC counters to stages.  Checked on two plans that take different kernels (polynomial at a generic shape; 4096-point fast path
    columns) and through the solver graph's replay (second call)."""
    from sfft_amd.plan import Plan
    from sfft_amd.utils.synthetic import make_pair
    for shape, expect in (((192, 160), {"fwd_rows": "rows_r2c", "greek_g2": "greek_g2", "fill": "fill_system", "solve": "chol_dataflow",
                                        "construct": "vconv_mixed", "inverse": "rows_c2r_diff"}),
                          ((64, 4096), {"fwd_rows": "rows_r2c_4096", "solve": "chol_dataflow", "inverse": "rows_c2r_diff_4096"})):
        pair = make_pair(*shape, seed=5, mask=True, density=400.0)
        g = {k: _to(dev, v) for k, v in pair.items()}
        plan = Plan(shape[0], shape[1], 3, 2, 1, True, device=dev.index)
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            plan.subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"])          # (captures the solver graph)
            plan.set_timing(True)
            plan.subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"])          # (replays it)
            ms, names = plan.stage_ms(), plan.stage_kernels()
        plan.set_timing(False)
        for st, frag in expect.items():
            assert ms[st] > 0.0, (shape, st)
            assert any(frag in k for k in names[st]), (shape, st, names[st])
        assert any("chol_back" in k or "scatter_solution" in k for k in names["solve"]), names["solve"]
        assert all(len(set(v)) == len(v) for v in names.values())            # each kernel once per stage
        plan.close()
