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


def test_pair_major_stage_lines_rows_writer_and_column_reader_agree():
    """Round 6: the stage planes of the 4096^2 solve pass store every 128-byte line pair-major.  What rows_r2c_4096 (pm = 1) writes -- one DPP
    quad exchange per store pair -- covers every element of its row pair exactly once at the offset of the layout's definition, and the loads
    of cols_fwd_weighted_4096_z (lane = 2 j + c, 16 loads per lane) hit exactly the (row, column) elements its transform needs."""
    pstride = 4096 * 4
    for l0 in (0, 2, 4094):
        st = M.rows_pm_store_offsets(l0, pstride)
        assert len(st) == 2 * 2052                                   # two rows x the 513 quads of the half spectrum (padding columns included)
        for off, (row, col) in st.items():
            assert off == M.pair_major_offset(row, col, pstride)
    for cp in (0, 1, 6, 1024):
        seen = set()
        for tid in range(512):
            for r in range(16):
                off, (row, col) = M.cols_z_load_offset(tid, r, cp, pstride)
                assert off == M.pair_major_offset(row, col, pstride)
                seen.add((row, col))
        assert seen == {(l, 2 * cp + c) for l in range(4096) for c in (0, 1)}


def test_h2048_network_and_real_row_untangle():
    """The 2048-point network of scripts/micro/fft_h2048.hpp (8 points per thread, three exchanges; measured in round 6, not the product path):
    its outputs are the DFT at the claimed indices, and the in-thread untangle turns them into the half spectrum of the real row."""
    rng = np.random.default_rng(0)
    z = rng.standard_normal(2048) + 1j * rng.standard_normal(2048)
    A, B, C, Cp = M.h2048_forward(z)
    Z = np.fft.fft(z)
    for s4 in range(4):
        assert np.abs(A[s4] - Z[C + 512 * s4]).max() <= 1e-12 and np.abs(B[s4] - Z[Cp + 512 * s4]).max() <= 1e-12
    x = rng.standard_normal(4096)
    A, B, C, Cp = M.h2048_forward(x[0::2] + 1j * x[1::2])
    tw = np.exp(-2j * np.pi * np.arange(4096) / 4096)
    X = np.zeros(2049, complex)
    t = np.arange(256)
    for s4 in range(4):
        k = t + 512 * s4
        Xk, Xp = M.h2048_untangle(A[s4], B[3 - s4], tw[k])
        X[k[1:]] = Xk[1:]
        X[2048 - k[1:]] = Xp[1:]
    A0, B0 = A[:, 0], B[:, 0]                                       # thread 0: the self-partnered combos 0 and 256
    X[0], X[2048] = M.h2048_untangle(A0[0], A0[0], tw[0])
    X[512], X[1536] = M.h2048_untangle(A0[1], A0[3], tw[512])
    X[1024], _ = M.h2048_untangle(A0[2], A0[2], tw[1024])
    X[256], X[1792] = M.h2048_untangle(B0[0], B0[3], tw[256])
    X[768], X[1280] = M.h2048_untangle(B0[1], B0[2], tw[768])
    assert np.abs(X - np.fft.rfft(x)).max() <= 1e-11
