"""CPU: the numpy oracle against golden vectors produced by the reference's own Numpy backend."""
import numpy as np
import pytest

from oracle import sfft_oracle as O
from _golden import golden_names, load_golden, packet_roles, rms, rel_rms_err

NAMES = golden_names()


def _params(meta):
    return O.SSC(meta["N0"], meta["N1"], meta["KerHW"], meta["DK"], meta["DB"], bool(meta["CPR"]))


@pytest.mark.parametrize("name", NAMES)
def test_linear_system_matches_reference(name):
    g = load_golden(name)
    p = _params(g["meta"])
    I, J, mI, mJ, _ = packet_roles(g)
    LHMAT, RHb = O.establish_system(mI, mJ, p)
    # element-wise, relative to the block maximum (<= 1e-11)
    assert np.max(np.abs(LHMAT - g["LHMAT"])) <= 1e-11 * np.max(np.abs(g["LHMAT"]))
    assert np.max(np.abs(RHb - g["RHb"])) <= 1e-11 * np.max(np.abs(g["RHb"]))


@pytest.mark.parametrize("name", NAMES)
def test_apply_given_reference_solution(name):
    """Apply-only gate: DIFF from the reference's Solution, pixel RMS error <= 1e-10 * RMS(J)."""
    g = load_golden(name)
    p = _params(g["meta"])
    I, J, mI, mJ, nm = packet_roles(g)
    DIFF = O.ESS(I, J, p, SFFTSolution=g["Solution"], Subtract=True)[1]
    if nm is not None:
        DIFF[nm] = np.nan
    if g["meta"]["ForceConv"] == "SCI":
        DIFF = -DIFF
    assert rms(DIFF - g["DIFF"]) <= 1e-10 * rms(J)


@pytest.mark.parametrize("name", NAMES)
def test_end_to_end(name):
    """End-to-end gate: DIFF pixel-RMS error <= 1e-6 * RMS(DIFF_ref)."""
    g = load_golden(name)
    m = g["meta"]
    Solution, DIFF = O.CP_arrays(g["REF"], g["SCI"], g["mREF"], g["mSCI"], m["ForceConv"], m["KerHW"],
                                 m["DK"], m["DB"], bool(m["CPR"]))
    assert np.array_equal(np.isnan(DIFF), np.isnan(g["DIFF"]))
    assert rel_rms_err(DIFF, g["DIFF"]) <= 1e-6
    assert Solution.shape == g["Solution"].shape


@pytest.mark.parametrize("name", [n for n in NAMES if "contam" in n or n == "c96x80_w8_k2b2_cpr"])
def test_contamination_mask_matches_reference(name):
    """GSS's ContamMask_I -> ContamMask_CI branch (SFFTSubtract.py:907-921) against the reference's own output: identical
    masks on every pixel whose convolved-mask value (stored from the reference's third ESS call) is not on the threshold."""
    g = load_golden(name)
    p = _params(g["meta"])
    I, J, mI, mJ, _ = packet_roles(g)
    D_ref, cm_ref = g["ContamD"], g["ContamMask_CI"]
    tsol = g["Solution"].copy()
    tsol[-p["Fpq"]:] = 0.0
    D = O.ESS(g["ContamMask_I"].astype(np.float64), np.zeros(J.shape), p, tsol, True)[1]
    assert np.max(np.abs(D - D_ref)) <= 1e-10 * np.max(np.abs(D_ref))
    clear = np.abs(D_ref + 0.001) >= 1e-9
    assert np.array_equal((D < -0.001)[clear], cm_ref[clear]) and clear.mean() > 0.99
    cmask = O.GSS(I, J, mI, mJ, p, ContamMask_I=g["ContamMask_I"])[2]
    clear = np.abs(D_ref + 0.001) >= 1e-6 * max(1.0, float(np.max(np.abs(D_ref))))
    assert np.array_equal(cmask[clear], cm_ref[clear]) and clear.mean() > 0.98


@pytest.mark.parametrize("name", [n for n in NAMES if "48x40" in n or "45x35" in n])
def test_construct_fdiff_literal_equals_matmul_form(name):
    g = load_golden(name)
    p = _params(g["meta"])
    I, J, _, _, _ = packet_roles(g)
    d1 = O.ESS(I, J, p, SFFTSolution=g["Solution"], Subtract=True, literal=True)[1]
    d2 = O.ESS(I, J, p, SFFTSolution=g["Solution"], Subtract=True, literal=False)[1]
    assert rms(d1 - d2) <= 1e-11 * rms(J)


def test_param_validation():
    with pytest.raises(Exception, match="KerPolyOrder should be 0/1/2/3"):
        O.SSC(64, 64, 2, 4, 0)
    with pytest.raises(Exception, match="BGPolyOrder should be 0/1/2/3"):
        O.SSC(64, 64, 2, 0, -1)
    p = O.SSC(4096, 4096, 8, 2, 2, True)
    assert (p["NEQ"], p["NEQ_FSfree"], p["Fijab"], p["Fab"]) == (1740, 1735, 1734, 289)


# ---------------------------------------------------------------------------------------------------
# B-spline form (oracle/bspline_oracle.py) against vectors from the reference's dev-version Numpy backend
# ---------------------------------------------------------------------------------------------------
from oracle import bspline_oracle as BO
from _golden import bspline_golden_names, load_bspline_golden

BS_NAMES = bspline_golden_names()


def _bs_setup(m):
    basis = BO.make_basis(m["N0"], m["N1"], m["KerSpType"], m["KerSpDegree"], m["KerIntKnotX"], m["KerIntKnotY"],
                          m["BkgSpType"], m["BkgSpDegree"], m["BkgIntKnotX"], m["BkgIntKnotY"])
    return basis, BO.SSC(m["N0"], m["N1"], m["w"], basis, bool(m["CPR"]))


@pytest.mark.parametrize("name", BS_NAMES)
def test_bspline_oracle_matches_reference(name):
    g = load_bspline_golden(name)
    basis, p = _bs_setup(g["meta"])
    assert (p["NEQ"], p["Fij"], p["Fpq"]) == (g["meta"]["NEQ"], g["meta"]["Fij"], g["meta"]["Fpq"])
    LHMAT, RHb = BO.establish_system(g["mREF"], g["mSCI"], p, basis)
    assert np.max(np.abs(LHMAT - g["LHMAT"])) <= 1e-11 * np.max(np.abs(g["LHMAT"]))
    assert np.max(np.abs(RHb - g["RHb"])) <= 1e-11 * np.max(np.abs(g["RHb"]))
    # apply-only with the reference's solution
    D = BO.ESS(g["REF"], g["SCI"], p, basis, SFFTSolution=g["Solution"], Subtract=True)[1]
    assert rms(D - g["DIFF"]) <= 1e-10 * rms(g["SCI"])
    # end to end, including the tied-scaling (TweakLS sum) rule
    sol, D2 = BO.GSS(g["REF"], g["SCI"], g["mREF"], g["mSCI"], p, basis)
    assert rel_rms_err(D2, g["DIFF"]) <= 1e-6
    if bool(g["meta"]["CPR"]) and g["meta"]["KerSpType"] == "B-Spline":
        Fab = p["Fab"]
        ij00 = np.arange(p["w0"] * p["L1"] + p["w1"], p["Fijab"], Fab)
        assert np.all(sol[ij00] == sol[ij00[0]])          # tied scaling: all centre coefficients equal
