"""Post-subtraction classes of the B-spline module against vectors made by the reference's own code
(tests/golden/make_golden_bspline_post.py): solution decoding and matching-kernel realisation are host numpy -> CPU tests."""
import numpy as np
import pytest

from sfft_amd.BSplineSFFT import Read_SFFTSolution, BSpline_MatchingKernel, ConvKernel_Convertion
from sfft_amd.utils import minifits
from _golden import GOLDEN_DIR

import os
G = np.load(os.path.join(GOLDEN_DIR, "bspline_post_cases.npz"), allow_pickle=False)
NCASE = len([k for k in G.files if k.endswith("_meta")])


def _case(k):
    m = eval(str(G["c%d_meta" % k][0]), {"__builtins__": {}}, {"dict": dict})
    m["L"] = 2 * m["w"] + 1
    return m


def _reader_args(m):
    return dict(KerSpType=m["KerSpType"], N0=m["N0"], N1=m["N1"], DK=m["DK"], L0=m["L"], L1=m["L"], Fi=m["Fi"], Fj=m["Fj"],
                Fpq=m["Fpq"], SEPARATE_SCALING=m["SEPARATE_SCALING"], ScaSpType=m["ScaSpType"], DS=m["DS"], ScaFi=m["ScaFi"], ScaFj=m["ScaFj"])


@pytest.mark.parametrize("k", range(NCASE))
def test_read_solution_matches_reference(k):
    m = _case(k)
    kd, sd = Read_SFFTSolution().FromArray(Solution=G["c%d_sol" % k], **_reader_args(m))
    keys = [tuple(t) for t in G["c%d_kerkeys" % k]]
    assert list(kd.keys()) == keys
    ref = G["c%d_kerdict" % k]
    mine = np.array([kd[t] for t in keys])
    assert np.array_equal(np.isnan(mine), np.isnan(ref))
    assert np.array_equal(np.nan_to_num(mine), np.nan_to_num(ref))
    if "c%d_scakeys" % k in G.files:
        ref_sca = dict(zip([tuple(t) for t in G["c%d_scakeys" % k]], G["c%d_scadict" % k]))
        assert sd is not None and len(sd) > 0
        for t, v in sd.items():
            assert ref_sca[t] == v
    else:
        assert sd is None


@pytest.mark.parametrize("k", range(NCASE))
def test_matching_kernel_matches_reference(k):
    m = _case(k)
    ks = BSpline_MatchingKernel(G["c%d_xy" % k], VERBOSE_LEVEL=0).FromArray(
        Solution=G["c%d_sol" % k], KerIntKnotX=m["KerIntKnotX"], KerIntKnotY=m["KerIntKnotY"], ScaIntKnotX=m["ScaIntKnotX"],
        ScaIntKnotY=m["ScaIntKnotY"], **_reader_args(m))
    ref = G["c%d_kerstack" % k]
    assert ks.shape == ref.shape
    assert np.abs(ks - ref).max() <= 1e-13 * np.abs(ref).max()


def test_from_fits_reads_what_bsp_writes(tmp_path):
    """FromFITS through the keyword set BSpline_Packet.BSP stores next to the solution (BSplineSFFT.py:4282-4351)."""
    k = 3          # B-spline kernel with one knot per axis, polynomial scaling of degree 1
    m = _case(k)
    sol = G["c%d_sol" % k]
    cards = []
    kw = [("KERHW", m["w"]), ("KSPTYPE", m["KerSpType"]), ("KSPDEG", m["DK"]), ("NKIKX", len(m["KerIntKnotX"]))] + \
         [("KIKX%d" % i, v) for i, v in enumerate(m["KerIntKnotX"])] + [("NKIKY", len(m["KerIntKnotY"]))] + \
         [("KIKY%d" % i, v) for i, v in enumerate(m["KerIntKnotY"])] + \
         [("SEPSCA", "True"), ("SSPTYPE", m["ScaSpType"]), ("SSPDEG", m["DS"]), ("NSIKX", 0), ("NSIKY", 0),
          ("N0", m["N0"]), ("N1", m["N1"]), ("DK", m["DK"]), ("L0", m["L"]), ("L1", m["L"]), ("FI", m["Fi"]), ("FJ", m["Fj"]),
          ("FPQ", m["Fpq"]), ("SCAFI", -1), ("SCAFJ", -1)]
    for key, v in kw:
        minifits.set_card(cards, key, v, "SFFT")
    path = str(tmp_path / "sol.fits")
    minifits.writeto(path, np.ascontiguousarray(sol.reshape((-1, 1)).T), cards)
    kd, sd = Read_SFFTSolution().FromFITS(path)
    ref = G["c%d_kerdict" % k]
    assert np.array_equal(np.nan_to_num(np.array(list(kd.values()))), np.nan_to_num(ref))
    ks = BSpline_MatchingKernel(G["c%d_xy" % k], VERBOSE_LEVEL=0).FromFITS(path)
    assert np.abs(ks - G["c%d_kerstack" % k]).max() <= 1e-13 * np.abs(G["c%d_kerstack" % k]).max()


def test_csz_roundtrip_and_lost_weight():
    rng = np.random.default_rng(3)
    K = rng.normal(size=(7, 5))
    big = ConvKernel_Convertion.CSZ(K, 32, 24)
    assert big.shape == (32, 24) and big[0, 0] == K[3, 2] and big[-3, -2] == K[0, 0]
    back, lost = ConvKernel_Convertion.iCSZ(big, 7, 5)
    assert np.array_equal(back, K) and abs(lost) < 1e-15
