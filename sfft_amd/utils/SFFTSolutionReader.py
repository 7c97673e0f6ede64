"""Decoding of the Solution vector -- mirror of sfft/utils/SFFTSolutionReader.py.

Solution = [a_ijab (ij-major, ab row-major), b_pq].  ac_ijab = a_ijab / (N0*N1) are the coefficients of the matching
kernel on the "modified delta basis" (K_ab = delta(a,b) - delta(0,0), centre coefficient = kernel sum) times x^i y^j in
ScaledFortranCoor (SFFTSolutionReader.py:14-40).  Host-side numpy like the reference: these are a few kilobytes.
"""
from copy import deepcopy

import numpy as np

from . import minifits

try:  # pragma: no cover - astropy is absent from the target image
    from astropy.io import fits as _afits
except Exception:
    _afits = None

__all__ = ["Read_SFFTSolution", "SVKDict_ST2SFFT", "SVKDict_SFFT2ST", "Realize_MatchingKernel", "Realize_FluxScaling"]


def _ref_ij(DK):
    return [(i, j) for i in range(DK + 1) for j in range(DK + 1 - i)]


def _read_solution_fits(FITS_Solution):
    if _afits is not None:
        phr = _afits.getheader(FITS_Solution, ext=0)
        data = _afits.getdata(FITS_Solution, ext=0)
    else:
        data, cards = minifits.getdata(FITS_Solution)
        phr = minifits.header_dict(cards)
    return (np.asarray(data, dtype=np.float64)[0], int(phr['N0']), int(phr['N1']), int(phr['L0']), int(phr['L1']),
            int(phr['DK']), int(phr['FPQ']))


class Read_SFFTSolution:
    def FromArray(self, Solution, N0, N1, L0, L1, DK, Fpq):
        """Sfft_dict[(i, j)][a + w0, b + w1] = ac_ijab  (SFFTSolutionReader.py:44-74)."""
        pairs = _ref_ij(DK)
        Fab = L0 * L1
        a_ijab = np.asarray(Solution, dtype=np.float64)[:-Fpq]      # drop differential background
        ac = (a_ijab / (N0 * N1)).reshape(len(pairs), L0, L1)       # ab is row-major over (a, b)
        return {ij: ac[k].astype(float).copy() for k, ij in enumerate(pairs)}

    def FromFITS(self, FITS_Solution):
        Solution, N0, N1, L0, L1, DK, Fpq = _read_solution_fits(FITS_Solution)
        return self.FromArray(Solution=Solution, N0=N0, N1=N1, L0=L0, L1=L1, DK=DK, Fpq=Fpq)


class SVKDict_ST2SFFT:
    @staticmethod
    def convert(DKx, DKy, Standard_dict):
        """Standard (Cartesian delta) basis -> SFFT basis: the centre entry becomes the kernel sum (:89-100)."""
        L0, L1 = Standard_dict[(0, 0)].shape
        w0, w1 = int((L0 - 1) / 2), int((L1 - 1) / 2)
        Sfft_dict = deepcopy(Standard_dict)
        for i in range(DKx + 1):
            for j in range(DKy + 1 - i):
                Sfft_dict[(i, j)][w0, w1] = np.sum(Standard_dict[(i, j)])
        return Sfft_dict


class SVKDict_SFFT2ST:
    @staticmethod
    def convert(DKx, DKy, Sfft_dict):
        """SFFT basis -> standard basis: centre pixel = 2 * ac_ij00 - sum_ab ac_ijab (:102-114)."""
        L0, L1 = Sfft_dict[(0, 0)].shape
        w0, w1 = int((L0 - 1) / 2), int((L1 - 1) / 2)
        Standard_dict = deepcopy(Sfft_dict)
        for i in range(DKx + 1):
            for j in range(DKy + 1 - i):
                Standard_dict[(i, j)][w0, w1] = 2 * Sfft_dict[(i, j)][w0, w1] - np.sum(Sfft_dict[(i, j)])
        return Standard_dict


def _scaled(XY_q, N0, N1):
    sXY_q = np.asarray(XY_q).astype(float)     # FortranCoor -> ScaledFortranCoor
    sXY_q[:, 0] /= N0
    sXY_q[:, 1] /= N1
    return sXY_q


class Realize_MatchingKernel:
    def __init__(self, XY_q):
        self.XY_q = XY_q

    def FromArray(self, Solution, N0, N1, L0, L1, DK, Fpq):
        """Matching kernels in the standard basis at the requested coordinates: (Num_request, L0, L1) (:121-138)."""
        sXY_q = _scaled(self.XY_q, N0, N1)
        Sfft_dict = Read_SFFTSolution().FromArray(Solution=Solution, N0=N0, N1=N1, L0=L0, L1=L1, DK=DK, Fpq=Fpq)
        Standard_dict = SVKDict_SFFT2ST.convert(DKx=DK, DKy=DK, Sfft_dict=Sfft_dict)
        pairs = _ref_ij(DK)
        B = np.array([sXY_q[:, 0] ** i * sXY_q[:, 1] ** j for i, j in pairs])
        SSS = np.array([Standard_dict[ij] for ij in pairs])
        return np.tensordot(B, SSS, (0, 0))

    def FromFITS(self, FITS_Solution):
        Solution, N0, N1, L0, L1, DK, Fpq = _read_solution_fits(FITS_Solution)
        return self.FromArray(Solution=Solution, N0=N0, N1=N1, L0=L0, L1=L1, DK=DK, Fpq=Fpq)


class Realize_FluxScaling:
    def __init__(self, XY_q):
        self.XY_q = XY_q

    def FromArray(self, Solution, N0, N1, L0, L1, DK, Fpq):
        """Flux scaling (kernel sum) Ac_xy00 = sum_ij ac_ij00 x^i y^j at the requested coordinates (:158-183)."""
        w0, w1 = int((L0 - 1) / 2), int((L1 - 1) / 2)
        sXY_q = _scaled(self.XY_q, N0, N1)
        Sfft_dict = Read_SFFTSolution().FromArray(Solution=Solution, N0=N0, N1=N1, L0=L0, L1=L1, DK=DK, Fpq=Fpq)
        out = np.zeros(np.asarray(self.XY_q).shape[0]).astype(float)
        for (i, j) in _ref_ij(DK):
            out += Sfft_dict[(i, j)][w0, w1] * sXY_q[:, 0] ** i * sXY_q[:, 1] ** j
        return out

    def FromFITS(self, FITS_Solution):
        Solution, N0, N1, L0, L1, DK, Fpq = _read_solution_fits(FITS_Solution)
        return self.FromArray(Solution=Solution, N0=N0, N1=N1, L0=L0, L1=L1, DK=DK, Fpq=Fpq)
