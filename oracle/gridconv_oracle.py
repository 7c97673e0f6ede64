"""CPU restatement of BSpline_GridConvolve.GSVC_GPU (sfft/BSplineSFFT.py:4951-5006) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference runs the per-segment convolutions with cupyx.scipy.signal.convolve2d / fftconvolve; scipy.signal has the same
functions with the same semantics, so this is the reference's loop with `cp` replaced by `np` / scipy.

PARITY: neither CuPy nor astropy (needed by the reference's GSVC_CPU) is importable in the build container, so no
reference-generated vector exists for this function alone; the restatement leans on scipy.signal.convolve2d / fftconvolve
being the documented twins of the cupyx functions.  Since round 3 its FFT branch is pinned end to end through the reference's
NIRCam golden (oracle/nircam_chain.py, tests/test_nircam_chain.py: 81 tiles, 411 x 411 kernels).  Only tests/ may import this module."""
import numpy as np
from scipy.signal import convolve2d, fftconvolve


def gsvc(PixA_in, AllocatedL, KerStack, normalize_kernel=True, use_fft=False):
    N0, N1 = PixA_in.shape
    Nseg, L0, L1 = KerStack.shape
    w0, w1 = int((L0 - 1) / 2), int((L1 - 1) / 2)
    IBx, IBy = w0 + 1, w1 + 1
    if normalize_kernel:
        KerStack = KerStack / np.sum(KerStack, axis=(1, 2))[:, np.newaxis, np.newaxis]
    out = np.zeros((N0, N1), dtype=np.float64)
    for idx in range(Nseg):
        lX, lY = np.where(AllocatedL == idx)
        xs, xe, ys, ye = lX.min(), lX.max(), lY.min(), lY.max()
        xEs, xEe = max([0, xs - IBx]), min([N0 - 1, xe + IBx])
        yEs, yEe = max([0, ys - IBy]), min([N1 - 1, ye + IBy])
        mini = PixA_in[xEs: xEe + 1, yEs: yEe + 1]
        if use_fft:
            conv = fftconvolve(mini, KerStack[idx], mode='same')
        else:
            conv = convolve2d(mini, KerStack[idx], mode='same', boundary='fill', fillvalue=0.0)
        out[xs: xe + 1, ys: ye + 1] = conv[xs - xEs: (xs - xEs) + (xe + 1 - xs), ys - yEs: (ys - yEs) + (ye + 1 - ys)]
    return out


def tile_labels(N0, N1, TiHW):
    """The tiling of the reference's docstring example (BSplineSFFT.py:4883-4903): labels and tile centres (FortranCoor)."""
    TiN = 2 * TiHW + 1
    lab, XY = 0, []
    AllocatedL = np.zeros((N0, N1), dtype=int)
    for xs in np.arange(0, N0, TiN):
        xe = np.min([xs + TiN, N0])
        for ys in np.arange(0, N1, TiN):
            ye = np.min([ys + TiN, N1])
            AllocatedL[xs: xe, ys: ye] = lab
            XY.append([0.5 + xs + (xe - xs) / 2.0, 0.5 + ys + (ye - ys) / 2.0])
            lab += 1
    return AllocatedL, np.array(XY)
