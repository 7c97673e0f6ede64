"""The algebra behind the decimation step of the Greek stage-1 kernels (greek_g1_mfma4g<.., true>, greek_g1<.., true>) and behind the
tensor-basis mixed-domain apply (vconv_tensor), checked with numpy against the direct definitions.  No GPU, no library: these
pin the formulas the kernels implement, next to the oracle tests that pin their results."""
import numpy as np
import pytest


@pytest.mark.parametrize("N0,h", [(64, 16), (96, 8), (130, 5)])
def test_pruned_transform_over_half_the_rows(N0, h):
    """G[r] = sum_x H[x] W^(r x), W = exp(-2 pi i / N0), for |r| <= h: with Ye = H[x'] + H[x' + N0/2], Yo = H[x'] - H[x' + N0/2] the even lags
    are the same sum over x' < N0/2 of Ye and the odd lags of Yo (the twiddle at the partner row is (-1)^r times the one at x')."""
    rng = np.random.default_rng(N0)
    H = rng.normal(size=N0) + 1j * rng.normal(size=N0)
    x = np.arange(N0)
    xh = np.arange(N0 // 2)
    Ye, Yo = H[:N0 // 2] + H[N0 // 2:], H[:N0 // 2] - H[N0 // 2:]
    for r in range(-h, h + 1):
        direct = np.sum(H * np.exp(-2j * np.pi * r * x / N0))
        Y = Ye if r % 2 == 0 else Yo
        half = np.sum(Y * np.exp(-2j * np.pi * r * xh / N0))
        assert abs(direct - half) <= 1e-12 * np.sum(np.abs(H))
    # the four real sums the kernels keep per lag r >= 1 give both signs: G[+r] = (S1 - S2) + i (S3 + S4), G[-r] = (S1 + S2) + i (S4 - S3)
    for r in range(1, h + 1):
        Y = Ye if r % 2 == 0 else Yo
        w = np.exp(-2j * np.pi * r * xh / N0)
        S1, S2, S3, S4 = np.sum(w.real * Y.real), np.sum(w.imag * Y.imag), np.sum(w.imag * Y.real), np.sum(w.real * Y.imag)
        gp = np.sum(H * np.exp(-2j * np.pi * r * x / N0))
        gm = np.sum(H * np.exp(+2j * np.pi * r * x / N0))
        assert abs((S1 - S2) + 1j * (S3 + S4) - gp) <= 1e-12 * np.sum(np.abs(H))
        assert abs((S1 + S2) + 1j * (S4 - S3) - gm) <= 1e-12 * np.sum(np.abs(H))


@pytest.mark.parametrize("N0,w", [(48, 3), (40, 8)])
def test_inverse_column_transform_of_a_separable_term_in_closed_form(N0, w):
    """sum_l FI[l] W0^(l a) e^(+2 pi i l x / N0) = N0 * bx[x - a] * S[x - a] for FI = column-DFT(bx * S): the identity that lets the apply pass
    of any separable kernel term (polynomial or B-spline row factor bx) run as a (2w+1)-tap walk along the columns of the stage plane S."""
    rng = np.random.default_rng(w)
    bx = rng.normal(size=N0)                              # any row factor: cx^i or a B-spline basis function
    S = rng.normal(size=N0) + 1j * rng.normal(size=N0)    # one spectrum column of the stage plane row-DFT(I * by)
    FI = np.fft.fft(bx * S)
    l = np.arange(N0)
    for a in range(-w, w + 1):
        for xx in (0, 1, N0 // 2, N0 - 1):
            lhs = np.sum(FI * np.exp(-2j * np.pi * l * a / N0) * np.exp(2j * np.pi * l * xx / N0))
            rhs = N0 * bx[(xx - a) % N0] * S[(xx - a) % N0]
            assert abs(lhs - rhs) <= 1e-10 * N0 * np.max(np.abs(S)) * np.max(np.abs(bx))
