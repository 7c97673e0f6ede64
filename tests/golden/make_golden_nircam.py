#!/usr/bin/env python3
"""Pack the reference's NIRCam example (test/subtract_test_nircam) into one fixture: tests/golden/nircam_case.npz.

Build container only (reads /root/reference).  The fixture is DATA: the two 900 x 900 stamps, the two WebbPSF models, the two
noise maps, the mask, and the reference's own end product `4check/...sfftdiff.DeCorrelated.SNR.fits` -- the differential SNR map
its notebook `subtract4nircam.ipynb` (cells 4-14) derives from those inputs with B-spline SFFT (KerHW 11, B-spline kernel of
degree 2 with 2 x 2 internal knots, SEPARATE polynomial scaling of degree 2, Tikhonov regularisation with lambda = 3e-5 on 512
seeded points), tile-wise noise decorrelation and a 32-sample Monte-Carlo noise map.  It is the only artefact of the reference
in this checkout that went through SEPARATE-VARYING scaling, kernel regularisation and BSpline_GridConvolve
(sfft/BSplineSFFT.py:1350-2168, 3296-3700, 4870-5008), which have no CPU implementation to import.

Arrays are stored exactly as astropy's `fits.getdata` returns them ([NAXIS2][NAXIS1], float32; the mask as uint8), no source
text of the reference is stored.  tests/test_nircam_chain.py replays the notebook on them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from sfft_amd.utils import minifits  # noqa: E402

ROOT = "/root/reference/test/subtract_test_nircam"
REFNAME = "jw01324001001-01324-o001_t001_nircam_clear-f200w_i2d_stamp"
SCINAME = "jw02561001002-02561-o001_t003_nircam_clear-f200w_i2d_stamp"


def read(path):
    data, cards = minifits.getdata(path)
    hdr = minifits.header_dict(cards)
    assert hdr["BITPIX"] == -32 and data.dtype == np.float32, path
    return np.ascontiguousarray(data)


def main():
    out = dict(
        lREF=read("%s/input_data/%s.fits" % (ROOT, REFNAME)),
        lSCI=read("%s/input_data/%s.fits" % (ROOT, SCINAME)),
        PSF_lREF=read("%s/auxiliary/%s.WebbPSF.fits" % (ROOT, REFNAME)),
        PSF_lSCI=read("%s/auxiliary/%s.WebbPSF.fits" % (ROOT, SCINAME)),
        Noise_lREF=read("%s/auxiliary/%s.noise.fits" % (ROOT, REFNAME)),
        Noise_lSCI=read("%s/auxiliary/%s.noise.fits" % (ROOT, SCINAME)),
        DCDIFF_SNR=read("%s/4check/%s.crossConvd.sfftdiff.DeCorrelated.SNR.fits" % (ROOT, SCINAME)),
    )
    mask = read("%s/auxiliary/%s.mask4sfft.fits" % (ROOT, SCINAME))
    assert set(np.unique(mask)) <= {0.0, 1.0}
    out["mask4sfft"] = mask.astype(np.uint8)
    # SkyLevel_Estimator.SLE is numpy only: run the reference's own module on the two stamps (cell 11 of the notebook does) and keep
    # what it returns, so that the restatement in oracle/nircam_chain.py is pinned on its own as well
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_sle", "/root/reference/sfft/utils/SkyLevelEstimator.py")
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    out["sle_lREF"] = np.array(R.SkyLevel_Estimator.SLE(PixA_obj=out["lREF"].T), dtype=np.float64)
    out["sle_lSCI"] = np.array(R.SkyLevel_Estimator.SLE(PixA_obj=out["lSCI"].T), dtype=np.float64)
    out["meta"] = np.array([repr(dict(refname=REFNAME, sciname=SCINAME, source="test/subtract_test_nircam",
                                      notebook="subtract4nircam.ipynb cells 4-14", layout="fits.getdata order [NAXIS2][NAXIS1]"))])
    path = os.path.join(HERE, "nircam_case.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
