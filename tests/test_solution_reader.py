"""CPU: Solution decoding against vectors from the reference's own SFFTSolutionReader."""
import os

import numpy as np

from sfft_amd.utils import SFFTSolutionReader as R
from sfft_amd.utils import minifits

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reader_cases.npz"))


def test_reader_matches_reference(tmp_path):
    k = 0
    while "case%d_meta" % k in Z.files:
        N0, N1, w, DK, DB = [int(v) for v in Z["case%d_meta" % k]]
        L = 2 * w + 1
        Fpq = (DB + 1) * (DB + 2) // 2
        sol, XY = Z["case%d_sol" % k], Z["case%d_xy" % k]
        d = R.Read_SFFTSolution().FromArray(sol, N0, N1, L, L, DK, Fpq)
        st = R.SVKDict_SFFT2ST.convert(DK, DK, d)
        back = R.SVKDict_ST2SFFT.convert(DK, DK, st)
        assert np.array_equal(np.array([d[ij] for ij in sorted(d)]), Z["case%d_sfft" % k])
        assert np.allclose(np.array([st[ij] for ij in sorted(st)]), Z["case%d_std" % k], rtol=1e-15, atol=0)
        assert np.allclose(np.array([back[ij] for ij in sorted(back)]), Z["case%d_back" % k], rtol=1e-13, atol=1e-18)
        assert np.allclose(R.Realize_MatchingKernel(XY).FromArray(sol, N0, N1, L, L, DK, Fpq), Z["case%d_kers" % k], rtol=1e-14, atol=1e-18)
        assert np.allclose(R.Realize_FluxScaling(XY).FromArray(sol, N0, N1, L, L, DK, Fpq), Z["case%d_fscal" % k], rtol=1e-14, atol=1e-18)
        # FITS variant through the FITS_Solution layout written by Customized_Packet.CP (CustomizedPacket.py:205-221)
        cards = []
        for key, v in (("N0", N0), ("N1", N1), ("DK", DK), ("DB", DB), ("L0", L), ("L1", L), ("FPQ", Fpq)):
            minifits.set_card(cards, key, v, "MeLOn: SFFT")
        path = str(tmp_path / ("sol%d.fits" % k))
        minifits.writeto(path, np.ascontiguousarray(sol.reshape((-1, 1)).T), cards)
        assert np.allclose(R.Realize_FluxScaling(XY).FromFITS(path), Z["case%d_fscal" % k], rtol=1e-14, atol=1e-18)
        k += 1
    assert k == 4
