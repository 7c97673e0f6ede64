"""The reference's NIRCam example (test/subtract_test_nircam/subtract4nircam.ipynb, cells 4-14) replayed against the golden the
reference ships, `4check/...sfftdiff.DeCorrelated.SNR.fits` (tests/golden/nircam_case.npz, made by make_golden_nircam.py).

This is the one artefact of the reference in this checkout that went through SEPARATE-VARYING scaling, Tikhonov kernel
regularisation and BSpline_GridConvolve (sfft/BSplineSFFT.py:1350-2168, 3296-3700, 4870-5008 -- CuPy only, no CPU code to
import), so it is what pins SURVEY 8f row N4:

  * CPU (`-m "not gpu"`): the restated oracle chain (oracle/nircam_chain.py over oracle/bspline_sv_oracle.py) reproduces the
    reference's SNR map.  Measured in the build container: relative RMS error 2.5e-8, largest pixel error 8.0e-7 on a map of RMS
    1.19 -- the rounding floor of the golden's float32 pixels (2^-24 of a value ~ 6e-8 relative).  Gate: 2e-7.
  * GPU (`-m gpu`): the same notebook through the HIP operators (BSpline_Packet.BSP with the notebook's exact settings on float32
    FITS files, BSpline_MatchingKernel.FromFITS, BSpline_DeCorrelation.BDC per tile, BSpline_GridConvolve.GSVC_GPU with the
    411 x 411 decorrelation kernels) reproduces the golden to the same gate, and its difference image agrees with the oracle's
    to the 1e-6 end-to-end gate of the other parity tests.

What it does not pin: B-spline *scaling* bases (the notebook's scaling is polynomial), B-spline backgrounds (constant here), and
WEIGHT_REGULARIZE other than uniform.
"""
import os

import numpy as np
import pytest

from _golden import GOLDEN_DIR, rms

SNR_GATE = 2e-7         # relative RMS error of the SNR map against the reference's float32 golden


@pytest.fixture(scope="module")
def case():
    z = np.load(os.path.join(GOLDEN_DIR, "nircam_case.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def oracle_run(case):
    from oracle import nircam_chain as NC
    return NC.run(case, workers=min(16, os.cpu_count() or 1))


def _golden_snr(case):
    return case["DCDIFF_SNR"].T.astype(np.float64)


def test_sky_level_estimator_matches_reference(case):
    """SkyLevel_Estimator.SLE restated (numpy-1.x scalar rules written out) against the reference's own module run on the two stamps."""
    from oracle import nircam_chain as NC
    for k in ("lREF", "lSCI"):
        lvl, sig = NC.sky_level_estimator(case[k].T)
        assert abs(lvl - case["sle_" + k][0]) <= 1e-9 * abs(case["sle_" + k][0])
        assert abs(sig - case["sle_" + k][1]) <= 1e-9 * case["sle_" + k][1]


def test_convolve_fft_fill_is_zero_padded_linear_convolution():
    from scipy.signal import fftconvolve
    from oracle import nircam_chain as NC
    rng = np.random.default_rng(3)
    img = rng.normal(size=(40, 33))
    img[5, 7] = np.nan
    ker = rng.uniform(0.1, 1.0, size=(9, 7))
    filled = np.nan_to_num(img, nan=0.0)
    out = NC.convolve_fft_fill(img, ker, True)
    assert np.abs(out - fftconvolve(filled, ker / ker.sum(), mode="same")).max() <= 1e-13
    out = NC.convolve_fft_fill(img, ker, False)
    assert np.abs(out - fftconvolve(filled, ker, mode="same")).max() <= 1e-12


def test_oracle_chain_reproduces_reference_snr_map(case, oracle_run):
    G, S = _golden_snr(case), oracle_run["SNR"]
    assert S.shape == G.shape == (900, 900)
    B = np.ones(G.shape, dtype=bool)
    B[11:-11, 11:-11] = False
    assert np.all(S[B] == 0.0) and np.all(G[B] == 0.0)             # the KerHW border the notebook zeroes
    err = rms(S - G) / rms(G)
    print("oracle chain vs reference golden: rel RMS %.3e, max abs %.3e (RMS of the map %.4f)" % (err, np.abs(S - G).max(), rms(G)))
    assert err <= SNR_GATE
    assert np.abs(S - G).max() <= 5e-6
    p_sol = oracle_run["Solution"]
    assert p_sol.shape == (25 * 23 * 23 + 1,)
    ij00 = np.arange(11 * 23 + 11, 25 * 529, 529)
    assert np.all(p_sol[ij00[6:]] == 0.0) and np.all(p_sol[ij00[:6]] != 0.0)      # 6 scaling terms, 19 place-holders


# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_chain_reproduces_reference_snr_map(case, oracle_run, tmp_path):
    import torch
    from oracle import nircam_chain as NC
    from sfft_amd.BSplineSFFT import BSpline_Packet, BSpline_MatchingKernel, BSpline_DeCorrelation, BSpline_GridConvolve
    from sfft_amd.utils import minifits
    from sfft_amd.utils.PureCupyFFTKits import PureCupy_FFTKits
    dev = torch.device("cuda", 0)
    O = oracle_run
    N0, N1 = O["REF"].shape

    # cell 4 on the GPU: the library's zero-filled FFT convolution gives the notebook's cross-convolved images (before their
    # float32 hand-off); the chain then continues from the float32 images, bit-identical for both replays
    tg = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    conv = PureCupy_FFTKits.FFT_CONVOLVE(tg(np.nan_to_num(case["lREF"].T.astype(np.float64), nan=0.0)), tg(case["PSF_lSCI"].T),
                                         PAD_FILL_VALUE=0., NAN_FILL_VALUE=0., NORMALIZE_KERNEL=True).cpu().numpy()
    ref64 = NC.convolve_fft_fill(case["lREF"].T, case["PSF_lSCI"].T, True)
    assert np.abs(conv - ref64).max() <= 1e-12 * np.abs(ref64).max()

    # cells 6-8: float32 FITS files in, BSpline_Packet.BSP with the notebook's settings
    def wfits(name, A):
        path = str(tmp_path / name)
        minifits.writeto(path, np.ascontiguousarray(A.T.astype(np.float32)), [])
        return path
    FITS_REF, FITS_SCI = wfits("ref.crossConvd.fits", O["REF"]), wfits("sci.crossConvd.fits", O["SCI"])
    FITS_mREF, FITS_mSCI = wfits("ref.crossConvd.masked.fits", O["mREF"]), wfits("sci.crossConvd.masked.fits", O["mSCI"])
    FITS_DIFF, FITS_Solution = str(tmp_path / "sci.sfftdiff.fits"), str(tmp_path / "sci.sfftsolution.fits")
    s = NC.notebook_settings(N0, N1)
    kw = {k: s[k] for k in ("ForceConv", "GKerHW", "KerSpType", "KerSpDegree", "KerIntKnotX", "KerIntKnotY", "SEPARATE_SCALING",
                            "ScaSpType", "ScaSpDegree", "ScaIntKnotX", "ScaIntKnotY", "BkgSpType", "BkgSpDegree", "BkgIntKnotX",
                            "BkgIntKnotY", "REGULARIZE_KERNEL", "IGNORE_LAPLACIAN_KERCENT", "XY_REGULARIZE", "WEIGHT_REGULARIZE",
                            "LAMBDA_REGULARIZE")}
    Solution, DIFF = BSpline_Packet.BSP(FITS_REF=FITS_REF, FITS_SCI=FITS_SCI, FITS_mREF=FITS_mREF, FITS_mSCI=FITS_mSCI,
                                        FITS_DIFF=FITS_DIFF, FITS_Solution=FITS_Solution, BACKEND_4SUBTRACT='Cupy',
                                        CUDA_DEVICE_4SUBTRACT='0', MAX_THREADS_PER_BLOCK=7, MINIMIZE_GPU_MEMORY_USAGE=True,
                                        VERBOSE_LEVEL=1, **kw)
    Solution = np.asarray(Solution.cpu() if hasattr(Solution, "cpu") else Solution)
    DIFF = np.asarray(DIFF.cpu() if hasattr(DIFF, "cpu") else DIFF)
    e_diff = rms(DIFF - O["DIFF"]) / rms(O["DIFF"])
    e_sol = np.linalg.norm(Solution - O["Solution"]) / np.linalg.norm(O["Solution"])
    print("HIP vs oracle: DIFF rel RMS %.3e, Solution rel L2 %.3e" % (e_diff, e_sol))
    assert e_diff <= 1e-6
    assert np.array_equal(Solution == 0.0, O["Solution"] == 0.0)            # the 19 place-holder scaling unknowns

    # cells 10-11: matching kernels from the solution FITS file, decorrelation kernel per tile
    MKerStack = BSpline_MatchingKernel(XY_q=O["XY_TiC"], VERBOSE_LEVEL=0).FromFITS(FITS_Solution=FITS_Solution)
    assert MKerStack.shape == (81, 23, 23)
    assert np.abs(MKerStack - O["MKerStack"]).max() <= 1e-6 * np.abs(O["MKerStack"]).max()
    bkgsig_lREF, bkgsig_lSCI = O["bkgsig"]
    PSF_lREF, PSF_lSCI = case["PSF_lREF"].T.astype(np.float64), case["PSF_lSCI"].T.astype(np.float64)
    DCKerStack = np.array([BSpline_DeCorrelation.BDC(MK_JLst=[PSF_lREF], SkySig_JLst=[bkgsig_lSCI], MK_ILst=[PSF_lSCI],
                                                     SkySig_ILst=[bkgsig_lREF], MK_Fin=MKer, KERatio=2.0, DENO_CLIP_RATIO=100000.0,
                                                     VERBOSE_LEVEL=0, CUDA_DEVICE=0) for MKer in MKerStack])
    assert DCKerStack.shape == (81, 411, 411)
    assert np.abs(DCKerStack - O["DCKerStack"]).max() <= 1e-6 * np.abs(O["DCKerStack"]).max()

    # cell 12: grid-wise decorrelation (the cell passes PixA_DIFF itself), KerHW border zeroed.  First on the oracle's inputs:
    # the banded direct-sum kernel against the reference's per-tile fftconvolve loop on identical data
    gc = BSpline_GridConvolve(PixA_obj=O["DIFF"], AllocatedL=O["AllocatedL"], KerStack=O["DCKerStack"], nan_fill_value=0.0,
                              use_fft=True, normalize_kernel=True).GSVC_GPU(CUDA_DEVICE='0', CLEAN_GPU_MEMORY=True, nproc=32)
    B = NC.boundary_mask(N0, N1)
    gc[B] = 0.
    assert np.abs(gc - O["DCDIFF"]).max() <= 1e-11 * np.abs(O["DCDIFF"]).max()
    DCDIFF = BSpline_GridConvolve(PixA_obj=DIFF, AllocatedL=O["AllocatedL"], KerStack=DCKerStack, nan_fill_value=0.0,
                                  use_fft=True, normalize_kernel=True).GSVC_GPU(CUDA_DEVICE='0', CLEAN_GPU_MEMORY=True, nproc=32)
    DCDIFF[B] = 0.

    # cell 14: Monte-Carlo noise map with the mean kernels of THIS replay, SNR, and the reference's golden
    NoiseD = NC.noise_map(case, MKerStack, DCKerStack, workers=min(16, os.cpu_count() or 1))
    SNR = DCDIFF / NoiseD
    G = _golden_snr(case)
    err = rms(SNR - G) / rms(G)
    print("HIP chain vs reference golden: rel RMS %.3e, max abs %.3e; vs oracle chain: %.3e" %
          (err, np.abs(SNR - G).max(), rms(SNR - O["SNR"]) / rms(G)))
    assert err <= SNR_GATE
    assert np.abs(SNR - G).max() <= 5e-6
