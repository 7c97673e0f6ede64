"""CPU oracle for the FFT utilities -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of sfft/utils/PureCupyFFTKits.py (KERNEL_CSZ :37-53, FFT_CONVOLVE :73-105) and
sfft/utils/PureCupyDeCorrelationCalculator.py (PCDC :46-126, which cannot run here: CuPy).  Pinned indirectly: PCDC with
REAL_OUTPUT at the size DCC picks equals DeCorrelation_Calculator.DCC, whose reference output is tests/golden/decorr_case.npz
(tests/golden/make_golden_decorr.py runs the reference DCC on the reference's own test inputs).
"""
import numpy as np


def kernel_csz(K, N0, N1, normalize=False):
    L0, L1 = K.shape
    W0, W1 = (L0 - 1) // 2, (L1 - 1) // 2
    K = K / K.sum() if normalize else K
    return np.roll(np.roll(np.pad(K, ((0, N0 - L0), (0, N1 - L1))), -W0, axis=0), -W1, axis=1)


def fft_convolve(img, K, pad_fill=0.0, nan_fill=0.0, normalize=False):
    N0, N1 = img.shape
    L0, L1 = K.shape
    W0, W1 = (L0 - 1) // 2, (L1 - 1) // 2
    E = np.pad(img, ((W0, W0), (W1, W1)), constant_values=pad_fill)
    if nan_fill is not None:
        E[np.isnan(E)] = nan_fill
    KI = kernel_csz(K, N0 + 2 * W0, N1 + 2 * W1, normalize)
    return np.fft.ifft2(np.fft.fft2(E) * np.fft.fft2(KI)).real[W0:-W0, W1:-W1]


def pcdc_fourier(NX, NY, KJ, sigJ, KI=(), sigI=(), MK=None, normalize=True):
    delta = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=float)
    f2 = lambda K: np.abs(np.fft.fft2(kernel_csz(delta if K is None else K, NX, NY))) ** 2
    den = sum(s ** 2 * f2(K) / len(KJ) ** 2 for K, s in zip(KJ, sigJ))
    fmk = f2(MK)
    den = den + sum(s ** 2 * f2(K) * fmk / len(KI) ** 2 for K, s in zip(KI, sigI))
    F = 1.0 / np.sqrt(den)
    return F / F[0, 0] if normalize else F
