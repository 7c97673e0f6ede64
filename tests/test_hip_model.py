"""CPU: the numpy model of the HIP kernels' formulas (tests/_hip_model.py) against numpy.fft and the oracle.
Guards the algebra the kernels rely on (the GPU parity tests guard the kernels themselves)."""
import numpy as np
import pytest

import _hip_model as M
from oracle import sfft_oracle as O
from _golden import load_golden, packet_roles, rms


@pytest.mark.parametrize("n", [8, 16, 32, 64, 128, 512])
def test_stockham_matches_numpy(n):
    rng = np.random.default_rng(n)
    x = rng.normal(size=(3, n)) + 1j * rng.normal(size=(3, n))
    assert np.allclose(M.lds_fft(x), np.fft.fft(x, axis=1), rtol=0, atol=1e-12 * n)


@pytest.mark.parametrize("n", [9, 35, 40, 45, 48, 96, 100])
def test_bluestein_matches_numpy(n):
    rng = np.random.default_rng(n)
    ax = M.Axis(n)
    x = np.zeros((2, ax.M), complex)
    x[:, :n] = rng.normal(size=(2, n)) + 1j * rng.normal(size=(2, n))
    got = M.lds_dft(x, ax)[:, :n]
    assert np.allclose(got, np.fft.fft(x[:, :n], axis=1), rtol=0, atol=1e-11 * n)


@pytest.mark.parametrize("shape", [(16, 16), (45, 35), (48, 40), (33, 64)])
def test_forward_and_inverse_planes(shape):
    rng = np.random.default_rng(5)
    N0, N1 = shape
    img = rng.normal(size=shape)
    ax0, ax1 = M.Axis(N0), M.Axis(N1)
    cx = ((np.arange(N0) + 1.0) / N0)[:, None]
    cy = ((np.arange(N1) + 1.0) / N1)[None, :]
    F = M.forward_plane(img, 2, 1, ax0, ax1)
    ref = np.fft.fft2(img * cx ** 2 * cy) / (N0 * N1)
    assert np.allclose(F, ref[:, :N1 // 2 + 1], rtol=0, atol=1e-13)
    # inverse: DIFF = J - B - IDFT_unnormalised(FD) with FD the half spectrum of a real image
    conv = rng.normal(size=shape)
    FD = (np.fft.fft2(conv) / (N0 * N1))[:, :N1 // 2 + 1]
    J = rng.normal(size=shape)
    bpq = [0.5, -1.0, 2.0]
    ref_pq = [(0, 0), (0, 1), (1, 0)]
    D = M.inverse_diff(FD, J, bpq, ref_pq, ax0, ax1)
    B = 0.5 - 1.0 * cy + 2.0 * cx
    assert np.allclose(D, J - B - conv, rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["c48x40_w2_k1b1_cpr", "c45x35_w2_k1b2_free"])
def test_model_system_solution_and_diff(name):
    g = load_golden(name)
    m = g["meta"]
    p = O.SSC(m["N0"], m["N1"], m["KerHW"], m["DK"], m["DB"], bool(m["CPR"]))
    T = O.index_tables(p)
    I, J, mI, mJ, nm = packet_roles(g)
    LH, rhs = M.build_system(mI, mJ, p, T)
    assert np.max(np.abs(LH - g["LHMAT"])) <= 1e-11 * np.max(np.abs(g["LHMAT"]))
    assert np.max(np.abs(rhs - g["RHb"])) <= 1e-11 * np.max(np.abs(g["RHb"]))
    idx = T["IDX_nFS"] if p["ConstPhotRatio"] else np.arange(p["NEQ"])
    x = M.bordered_cholesky_solve(LH[np.ix_(idx, idx)], rhs[idx])
    sol = np.zeros(p["NEQ"])
    sol[idx] = x
    ax0, ax1 = M.Axis(p["N0"]), M.Axis(p["N1"])
    specI = [M.forward_plane(I, i, j, ax0, ax1) for (i, j) in T["REF_ij"]]
    FD = M.construct_fd(specI, sol, p, ax0, ax1)
    DIFF = M.inverse_diff(FD, J, sol[p["Fijab"]:], [tuple(t) for t in T["REF_pq"]], ax0, ax1)
    if nm is not None:
        DIFF[nm] = np.nan
    if m["ForceConv"] == "SCI":
        DIFF = -DIFF
    assert rms(DIFF - g["DIFF"]) <= 1e-6 * rms(g["DIFF"])
