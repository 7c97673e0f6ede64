"""CPU replay of the reference's NIRCam example, test/subtract_test_nircam/subtract4nircam.ipynb cells 4-14 --
TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The notebook goes from two 900 x 900 NIRCam stamps (+ WebbPSF models, noise maps, a mask) to a differential SNR map that the
reference ships as `4check/...sfftdiff.DeCorrelated.SNR.fits`; tests/golden/nircam_case.npz holds those inputs and that map
(tests/golden/make_golden_nircam.py).  Every step is restated here in numpy / scipy with the cell or file:line it follows:

    cell 4   cross-convolution with astropy's convolve_fft(boundary='fill', nan_treatment='fill', fill_value=0,
             normalize_kernel=True), results written back into float32 FITS images           -> convolve_fft_fill, cross_convolve
    cell 6   zero-masking with mask4sfft                                                       -> masked_pair
    cell 7   BSP settings: KerHW 11, B-spline kernel degree 2 with knots 0.5 + N {1/3, 2/3}, SEPARATE polynomial scaling of
             degree 2, constant background, REGULARIZE_KERNEL with 512 points after np.random.seed(10086), lambda 3e-5
                                                                                               -> notebook_settings
    cell 8   BSpline_Packet.BSP (sfft/BSplineSFFT.py:3967-4356): SEPARATE-VARYING system, regularisation, solve, subtraction
                                                                                               -> bsp (oracle/bspline_sv_oracle.py)
    cell 10  81 tiles of 111 x 111 pixels (TiHW = 5 * KerHW)                                   -> gridconv_oracle.tile_labels
    cell 11  BSpline_MatchingKernel.FromFITS (BSplineSFFT.py:4555-4662), SkyLevel_Estimator.SLE
             (sfft/utils/SkyLevelEstimator.py:7-315), BSpline_DeCorrelation.BDC (BSplineSFFT.py:4755-4868) per tile
                                                                                               -> matching_kernels, sky_level_estimator, bdc
    cell 12  BSpline_GridConvolve(...).GSVC_GPU (BSplineSFFT.py:4870-5006; note the cell passes PixA_DIFF, not the copy whose
             border it zeroed), border of KerHW pixels set to 0                               -> gridconv_oracle.gsvc
    cell 14  32-sample Monte-Carlo noise propagation (np.random.seed(10086 + idx) / (20172 + idx)), SNR = DCDIFF / std
                                                                                               -> noise_map, snr_map

What this pins: the restated SEPARATE-VARYING / regularisation oracle (bspline_sv_oracle.py) and the grid convolution against an
artefact the REFERENCE produced (on an A100 with CuPy), through post-processing steps that are themselves pinned (matching
kernel and BDC bit-identically by tests/golden/bspline_post_cases.npz).  The golden is a float32 image and the chain contains
float32 hand-offs (FITS files), so agreement is limited to ~1e-6 of the map's RMS at best; tests/test_nircam_chain.py states the
measured figure.  Only tests/ may import this module.
"""
import numpy as np
import scipy.fft as sfft_
from scipy.interpolate import BSpline

GKERHW = 11
TILESIZE_RATIO = 5
MCNSAMP = 32


# ------------------------------------------------------------------------------------------------------------------
# cell 4: astropy.convolution.convolve_fft as the notebook calls it
# ------------------------------------------------------------------------------------------------------------------
def convolve_fft_fill(array, kernel, normalize_kernel=True, workers=8):
    """astropy.convolution.convolve_fft(array, kernel, boundary='fill', nan_treatment='fill', fill_value=0.0,
    normalize_kernel=...) restated from its published algorithm (astropy 5/6 `convolve.py: convolve_fft`; astropy is not
    vendored by the reference and not importable here): NaN / inf pixels of the image become 0; the kernel is divided by its
    sum (normalize_kernel=True) or temporarily normalised and rescaled afterwards (False); with boundary='fill' both arrays
    are centred in a zero-filled complex array of side 2^ceil(log2(max(array shape + kernel shape))), the kernel is
    ifftshift-ed, the product of the two transforms is inverted and the image's own window is cropped out, real part."""
    array = np.array(array, dtype=complex)
    kernel = np.array(kernel, dtype=complex)
    bad = np.isnan(array) | np.isinf(array)
    array[bad] = 0.0
    kernel[np.isnan(kernel) | np.isinf(kernel)] = 0
    if normalize_kernel:
        normalized_kernel = kernel / kernel.sum()
        kernel_scale = 1
    else:
        kernel_scale = kernel.sum()
        normalized_kernel = kernel / kernel_scale
    arrayshape, kernshape = array.shape, kernel.shape
    fsize = int(2 ** np.ceil(np.log2(np.max(np.array(arrayshape) + np.array(kernshape)))))
    newshape = (fsize, fsize)
    arrayslices, kernslices = [], []
    for newdimsize, arraydimsize, kerndimsize in zip(newshape, arrayshape, kernshape):
        center = newdimsize - (newdimsize + 1) // 2
        arrayslices.append(slice(center - arraydimsize // 2, center + (arraydimsize + 1) // 2))
        kernslices.append(slice(center - kerndimsize // 2, center + (kerndimsize + 1) // 2))
    arrayslices, kernslices = tuple(arrayslices), tuple(kernslices)
    bigarray = np.zeros(newshape, dtype=complex)
    bigarray[arrayslices] = array
    bigkernel = np.zeros(newshape, dtype=complex)
    bigkernel[kernslices] = normalized_kernel
    arrayfft = sfft_.fftn(bigarray, workers=workers)
    kernfft = sfft_.fftn(np.fft.ifftshift(bigkernel), workers=workers)
    fftmult = arrayfft * kernfft
    fftmult *= kernel_scale
    rifft = sfft_.ifftn(fftmult, workers=workers)
    return rifft[arrayslices].real


def cross_convolve(case, workers=8):
    """Cell 4: REF (x) PSF_SCI and SCI (x) PSF_REF.  The notebook assigns the float64 results into the data of the float32 input
    FITS files (`hdl[0].data[:, :] = ...`) and every later step reads those files, so the hand-off is float32.
    Returns (PixA_REF, PixA_SCI) as the `.T` views later cells obtain from fits.getdata, float32."""
    lREF, lSCI = case["lREF"].T, case["lSCI"].T
    PSF_lREF, PSF_lSCI = case["PSF_lREF"].T, case["PSF_lSCI"].T
    REF_convd = convolve_fft_fill(lREF, PSF_lSCI, True, workers).astype(np.float32)
    SCI_convd = convolve_fft_fill(lSCI, PSF_lREF, True, workers).astype(np.float32)
    return REF_convd, SCI_convd


def masked_pair(case, PixA_REF, PixA_SCI):
    """Cell 6: pixels outside mask4sfft set to 0 (float32 files again)."""
    SUBTMASK = case["mask4sfft"].T.astype(bool)
    mREF, mSCI = PixA_REF.copy(), PixA_SCI.copy()
    mREF[~SUBTMASK] = 0.0
    mSCI[~SUBTMASK] = 0.0
    return mREF, mSCI


def notebook_settings(N0, N1):
    """Cell 7, verbatim values.  np.random.seed / np.random.uniform are the legacy MT19937 stream, stable across numpy versions."""
    rs = np.random.RandomState(10086)
    XY_REGULARIZE = np.array([rs.uniform(10., N0 - 10., 512), rs.uniform(10., N1 - 10., 512)]).T
    return dict(ForceConv='REF', GKerHW=GKERHW, KerSpType='B-Spline', KerSpDegree=2,
                KerIntKnotX=[0.5 + N0 * 1 / 3, 0.5 + N0 * 2 / 3], KerIntKnotY=[0.5 + N1 * 1 / 3, 0.5 + N1 * 2 / 3],
                SEPARATE_SCALING=True, ScaSpType='Polynomial', ScaSpDegree=2, ScaIntKnotX=[], ScaIntKnotY=[],
                BkgSpType='Polynomial', BkgSpDegree=0, BkgIntKnotX=[], BkgIntKnotY=[],
                REGULARIZE_KERNEL=True, IGNORE_LAPLACIAN_KERCENT=True, XY_REGULARIZE=XY_REGULARIZE, WEIGHT_REGULARIZE=None,
                LAMBDA_REGULARIZE=3 * 1e-5)


# ------------------------------------------------------------------------------------------------------------------
# cell 8: BSpline_Packet.BSP through the restated SEPARATE-VARYING oracle
# ------------------------------------------------------------------------------------------------------------------
def bsp(PixA_REF, PixA_SCI, PixA_mREF, PixA_mSCI, s, workers=8, return_system=False):
    """BSplineSFFT.py:4108-4222 for ForceConv='REF' and NaN-free inputs (the cross-convolution filled the 5 NaN pixels of the
    reference stamp): float64 copies, (I, J) = (REF, SCI), GSS = solve on the masked pair + apply on the full pair."""
    from . import bspline_oracle as BO, bspline_sv_oracle as SV
    assert s['ForceConv'] == 'REF'
    I, J = np.ascontiguousarray(PixA_REF, np.float64), np.ascontiguousarray(PixA_SCI, np.float64)
    mI, mJ = np.ascontiguousarray(PixA_mREF, np.float64), np.ascontiguousarray(PixA_mSCI, np.float64)
    assert not (np.isnan(I).any() or np.isnan(J).any() or np.isnan(mI).any() or np.isnan(mJ).any())
    N0, N1 = I.shape
    basis = BO.make_basis(N0, N1, s['KerSpType'], s['KerSpDegree'], s['KerIntKnotX'], s['KerIntKnotY'],
                          s['BkgSpType'], s['BkgSpDegree'], s['BkgIntKnotX'], s['BkgIntKnotY'])
    Fij = len(basis['ker_pairs'])
    sca = SV.make_scaling_basis(N0, N1, Fij, s['ScaSpType'], s['ScaSpDegree'], s['ScaIntKnotX'], s['ScaIntKnotY'])
    p = SV.SSC(N0, N1, s['GKerHW'], basis, sca, 'SEPARATE-VARYING')
    kerspec = dict(KerSpType=s['KerSpType'], DK=s['KerSpDegree'], KerIntKnotX=s['KerIntKnotX'], KerIntKnotY=s['KerIntKnotY'])
    SST, CSST, DSST = SV.spatial_gram(p, kerspec, sca, s['XY_REGULARIZE'], s['WEIGHT_REGULARIZE'])
    iREG = SV.laplacian_ireg(p['w0'], p['w1'], s['IGNORE_LAPLACIAN_KERCENT'])
    LHMAT, RHb = SV.establish_system(mI, mJ, p, basis, sca, workers)
    # LHMAT += lambda * REGMAT block by block (REGMAT as a whole would be a second 1.4 GB matrix)
    Fab, c0, SCALE, lam = p['Fab'], p['w0'] * p['L1'] + p['w1'], p['SCALE'], s['LAMBDA_REGULARIZE']
    ir = iREG.astype(np.float64)
    for k in range(Fij):
        for k8 in range(Fij):
            blk = SCALE ** 2 * SST[k, k8] * ir
            blk[:, c0] = SCALE ** 2 * CSST[k, k8] * ir[:, c0]
            blk[c0, :] = SCALE ** 2 * CSST[k8, k] * ir[c0, :]
            blk[c0, c0] = SCALE ** 2 * DSST[k, k8] * ir[c0, c0]
            LHMAT[k * Fab:(k + 1) * Fab, k8 * Fab:(k8 + 1) * Fab] += lam * blk
    Solution = SV.solve_system(LHMAT, RHb, p)
    DIFF = SV.subtract(I, J, Solution, p, basis, sca, workers)
    if return_system:
        return Solution, DIFF, p, LHMAT, RHb
    return Solution, DIFF, p


# ------------------------------------------------------------------------------------------------------------------
# cell 11: matching kernels, sky sigma, decorrelation kernels
# ------------------------------------------------------------------------------------------------------------------
def _bspline_basis_req(N, IntKnot, k, ReqCoord):
    Knot = np.concatenate(([0.5] * (k + 1), IntKnot, [N + 0.5] * (k + 1))) / N
    Nc = len(IntKnot) + k + 1
    return np.array([BSpline(t=Knot, c=(np.arange(Nc) == idx).astype(float), k=k, extrapolate=False)(ReqCoord) for idx in range(Nc)])


def matching_kernels(Solution, s, N0, N1, XY_q):
    """BSpline_MatchingKernel.FromArray (BSplineSFFT.py:4561-4662) for a B-spline kernel with SEPARATE-VARYING polynomial
    scaling: Read_SFFTSolution (:4417-4523) splits ac = a / (N0 N1) into kernel stamps (centre = NaN) and the ScaFij
    scaling coefficients carried by the centre entries of the first ScaFij terms; the stamps are combined with the basis
    at the requested FortranCoor positions and the centre pixel becomes (scaling) - (sum of the off-centre pixels)."""
    assert s['KerSpType'] == 'B-Spline' and s['ScaSpType'] == 'Polynomial' and s['SEPARATE_SCALING'] and s['ScaSpDegree'] > 0
    w = s['GKerHW']
    L = 2 * w + 1
    DK, DS, Fpq = s['KerSpDegree'], s['ScaSpDegree'], 1
    Fi, Fj = len(s['KerIntKnotX']) + DK + 1, len(s['KerIntKnotY']) + DK + 1
    Fij = Fi * Fj
    sXY = np.array(XY_q, dtype=float)
    sXY[:, 0] /= N0
    sXY[:, 1] /= N1
    ac = (np.asarray(Solution, dtype=np.float64)[:-Fpq] / (N0 * N1)).reshape(Fij, L, L)
    sca_terms = [(i, j) for i in range(DS + 1) for j in range(DS + 1 - i)]
    ScaCOEFF = np.array([ac[k, w, w] for k in range(len(sca_terms))])
    KerCOEFF = ac.copy()
    KerCOEFF[:, w, w] = np.nan
    BX = _bspline_basis_req(N0, s['KerIntKnotX'], DK, sXY[:, 0])
    BY = _bspline_basis_req(N1, s['KerIntKnotY'], DK, sXY[:, 1])
    KerBASE = np.array([BX[i] * BY[j] for i in range(Fi) for j in range(Fj)])
    KerStack = np.tensordot(KerBASE, KerCOEFF, (0, 0))
    ScaBASE = np.array([sXY[:, 0] ** i * sXY[:, 1] ** j for i, j in sca_terms])
    KerCENT = np.matmul(ScaCOEFF.reshape((1, -1)), ScaBASE)[0]
    KerCENT -= np.nansum(KerStack, axis=(1, 2))
    KerStack[:, w, w] = KerCENT
    return KerStack


def sky_level_estimator(PixA_obj):
    """SkyLevel_Estimator.SLE (sfft/utils/SkyLevelEstimator.py:7-315): DAOPHOT's MMM on all finite pixels; returns
    (mode, sigma).  The notebook feeds float32 arrays and the reference ran under numpy 1.x, whose scalar rules differ from
    numpy 2 (python float x float32 scalar -> float64; float32 array (op) float64 scalar -> float32); the casts below are
    written out so that both numpy generations walk the same branches.  Defaults of the reference: no highbad, no readnoise,
    mxiter 50, minsky 20."""
    mxiter, minsky = 50, 20
    v = np.asarray(PixA_obj)
    sky = np.sort(v[v == v].ravel())
    nsky = len(sky)
    if nsky < minsky:
        return np.nan, -1.0
    nlast = nsky - 1
    f64 = np.float64
    skymid = 0.5 * f64(sky[int((nsky - 1) / 2)]) + 0.5 * f64(sky[int(nsky / 2)])
    cut1 = np.min([skymid - f64(sky[0]), f64(sky[nsky - 1]) - skymid])
    cut2 = skymid + cut1
    cut1 = skymid - cut1
    st = sky.dtype.type
    good = np.where((sky <= st(cut2)) & (sky >= st(cut1)))[0]
    if len(good) == 0:
        return 0.0, -1.0
    delta = sky[good] - st(skymid)
    sum_ = np.sum(delta.astype('float64'))
    sumsq = np.sum(delta.astype('float64') ** 2)
    maximm = int(np.max(good))
    minimm = int(np.min(good)) - 1
    skymed = 0.5 * f64(sky[int((minimm + maximm + 1) / 2)]) + 0.5 * f64(sky[int((minimm + maximm) / 2 + 1)])
    skymn = sum_ / (maximm - minimm)
    sigma = np.sqrt(sumsq / (maximm - minimm) - skymn ** 2)
    skymn = skymn + skymid
    skymod = 3. * skymed - 2. * skymn if skymed < skymn else skymn
    niter, clamp, old = 0, 1, 0
    redo = True
    while redo:
        niter += 1
        if niter > mxiter:
            return skymod, -1.0
        if maximm - minimm < minsky:
            return skymod, -1.0
        r = np.log10(float(maximm - minimm))
        r = np.max([2., (-0.1042 * r + 1.1695) * r + 0.8895])
        cut = r * sigma + 0.5 * np.abs(skymn - skymod)
        cut1, cut2 = skymod - cut, skymod + cut
        redo = False
        newmin = minimm
        tst_min = 1 if sky[newmin + 1] >= cut1 else 0
        done = 1 if (newmin == -1) and tst_min else 0
        if not done:
            skyind = newmin if newmin > 0 else 0
            if (sky[skyind] < cut1) and tst_min:
                done = 1
        if not done:
            istep = 1 - 2 * int(tst_min)
            while not done:
                newmin = newmin + istep
                if (newmin == -1) | (newmin == nlast):
                    done = 1
                if not done:
                    if (sky[newmin] <= cut1) and (sky[newmin + 1] >= cut1):
                        done = 1
            if tst_min:
                delta = sky[newmin + 1:minimm + 1] - st(skymid)
            else:
                delta = sky[minimm + 1:newmin + 1] - st(skymid)
            sum_ = sum_ - istep * f64(np.sum(delta))
            sumsq = sumsq - istep * f64(np.sum(delta ** 2))
            redo = True
            minimm = newmin
        newmax = maximm
        tst_max = 1 if sky[maximm] <= cut2 else 0
        done = 1 if (maximm == nlast) and tst_max else 0
        if not done:
            skyind = maximm + 1 if maximm + 1 < nlast else nlast
            if tst_max and (sky[skyind] > cut2):
                done = 1
        if not done:
            istep = -1 + 2 * int(tst_max)
            while not done:
                newmax = newmax + istep
                if (newmax == nlast) or (newmax == -1):
                    done = 1
                if not done:
                    if (sky[newmax] <= cut2) and (sky[newmax + 1] >= cut2):
                        done = 1
            if tst_max:
                delta = sky[maximm + 1:newmax + 1] - st(skymid)
            else:
                delta = sky[newmax + 1:maximm + 1] - st(skymid)
            sum_ = sum_ + istep * f64(np.sum(delta))
            sumsq = sumsq + istep * f64(np.sum(delta ** 2))
            redo = True
            maximm = newmax
        nsky = maximm - minimm
        if nsky < minsky:
            return skymod, -1.0
        skymn = sum_ / nsky
        var = sumsq / nsky - skymn ** 2
        if var < 0:
            var = 0
        sigma = float(np.sqrt(var))
        skymn = skymn + skymid
        center = (minimm + 1 + maximm) / 2.
        side = np.round(0.2 * (maximm - minimm)) / 2. + 0.25
        j = np.round(center - side)
        k = np.round(center + side)
        skymed = f64(np.sum(sky[int(j):int(k + 1)])) / (k - j + 1)
        dmod = 3. * skymed - 2. * skymn - skymod if skymed < skymn else skymn - skymod
        if dmod * old < 0:
            clamp = 0.5 * clamp
        skymod = skymod + clamp * dmod
        old = dmod
    return skymod, sigma


def _csz(K, N0, N1):
    """ConvKernel_Convertion.CSZ (BSplineSFFT.py:4725-4737): zero-pad to the image size and roll the centre to [0, 0]."""
    L0, L1 = K.shape
    out = np.zeros((N0, N1), dtype=float)
    out[:L0, :L1] = K
    return np.roll(np.roll(out, -((L0 - 1) // 2), axis=0), -((L1 - 1) // 2), axis=1)


def bdc(MK_JLst, SkySig_JLst, MK_ILst, SkySig_ILst, MK_Fin, KERatio=2.0, DENO_CLIP_RATIO=100000.0):
    """BSpline_DeCorrelation.BDC, image-subtraction mode with every kernel given (BSplineSFFT.py:4755-4868)."""
    import math
    NumI, NumJ = len(MK_ILst), len(MK_JLst)
    MK_Queue = list(MK_JLst) + [MK_Fin] + list(MK_ILst)
    L0_KDeCo = int(round(KERatio * np.max([MK.shape[0] for MK in MK_Queue])))
    L1_KDeCo = int(round(KERatio * np.max([MK.shape[1] for MK in MK_Queue])))
    L0_KDeCo += 1 if L0_KDeCo % 2 == 0 else 0
    L1_KDeCo += 1 if L1_KDeCo % 2 == 0 else 0
    N0 = 2 ** (math.ceil(np.log2(np.max([MK.shape[0] for MK in MK_Queue]))) + 1)
    N1 = 2 ** (math.ceil(np.log2(np.max([MK.shape[1] for MK in MK_Queue]))) + 1)

    def kft2(MK):
        kft = np.fft.fft2(_csz(MK, N0, N1))
        return (np.conj(kft) * kft).real
    kft2_Fin = kft2(MK_Fin)
    DeNo = 0.0
    for MKj, skysig in zip(MK_JLst, SkySig_JLst):
        DeNo = DeNo + (skysig ** 2 * kft2(MKj)) / NumJ ** 2
    for MKi, skysig in zip(MK_ILst, SkySig_ILst):
        DeNo = DeNo + (skysig ** 2 * kft2(MKi) * kft2_Fin) / NumI ** 2
    thresh = np.max(DeNo) / DENO_CLIP_RATIO
    DeNo[DeNo < thresh] = thresh
    DeCo = np.fft.ifft2(np.sqrt(1.0 / DeNo)).real
    back = np.roll(np.roll(DeCo, (L0_KDeCo - 1) // 2, axis=0), (L1_KDeCo - 1) // 2, axis=1)     # iCSZ (:4739-4753)
    KDeCo = back[:L0_KDeCo, :L1_KDeCo]
    return KDeCo / np.sum(KDeCo)


def decorrelation_kernels(case, MKerStack, bdc_fn=None):
    """Cell 11: sky sigmas of the UNCONVOLVED stamps (float32 arrays as read from FITS, NaNs ignored by SLE) and one BDC per tile:
    J side = the science image (convolved with the reference's PSF), I side = the reference (convolved with the science PSF,
    then with the matching kernel)."""
    bdc_fn = bdc if bdc_fn is None else bdc_fn
    PSF_lREF, PSF_lSCI = case["PSF_lREF"].T, case["PSF_lSCI"].T
    bkgsig_lREF = sky_level_estimator(case["lREF"].T)[1]
    bkgsig_lSCI = sky_level_estimator(case["lSCI"].T)[1]
    out = [bdc_fn(MK_JLst=[PSF_lREF], SkySig_JLst=[bkgsig_lSCI], MK_ILst=[PSF_lSCI], SkySig_ILst=[bkgsig_lREF], MK_Fin=MKer,
                  KERatio=2.0, DENO_CLIP_RATIO=100000.0) for MKer in MKerStack]
    return np.array(out), (bkgsig_lREF, bkgsig_lSCI)


# ------------------------------------------------------------------------------------------------------------------
# cells 12 and 14
# ------------------------------------------------------------------------------------------------------------------
def boundary_mask(N0, N1):
    m = np.ones((N0, N1)).astype(bool)
    m[GKERHW:-GKERHW, GKERHW:-GKERHW] = False
    return m


def multi_convolve_noise(PixA_Noise, ConvKerSeq, KerNormalizeSeq, RANDOM_SEED, workers=8):
    """Cell 14's MultiConvolveNoise: sample idx draws N(0, 1) x noise map after np.random.seed(RANDOM_SEED + idx) (legacy
    MT19937 stream: reproducible) and goes through the convolution sequence."""
    out = []
    for idx in range(MCNSAMP):
        rs = np.random.RandomState(RANDOM_SEED + idx)
        S = rs.normal(0, 1, PixA_Noise.shape) * PixA_Noise
        for ConvKer, KerNormalize in zip(ConvKerSeq, KerNormalizeSeq):
            S = convolve_fft_fill(S, ConvKer, KerNormalize, workers)
        out.append(S)
    return np.array(out)


def noise_map(case, MKerStack, DCKerStack, workers=8):
    """Cell 14: propagated noise of the decorrelated difference with the MEAN decorrelation and matching kernels."""
    PSF_lREF, PSF_lSCI = case["PSF_lREF"].T, case["PSF_lSCI"].T
    DCKerMean = np.mean(DCKerStack.reshape(DCKerStack.shape[0], -1), axis=0).reshape(DCKerStack.shape[1:])
    MKerMean = np.mean(MKerStack.reshape(MKerStack.shape[0], -1), axis=0).reshape(MKerStack.shape[1:])
    S = multi_convolve_noise(case["Noise_lSCI"].T, [PSF_lREF, DCKerMean], [True, True], 10086, workers)
    R = multi_convolve_noise(case["Noise_lREF"].T, [PSF_lSCI, MKerMean, DCKerMean], [True, False, True], 2 * 10086, workers)
    return np.std(S - R, axis=0)


def run(case, workers=8, bsp_fn=None, gsvc_fn=None, bdc_fn=None, verbose=False):
    """The whole replay.  bsp_fn(REF, SCI, mREF, mSCI, settings) -> (Solution, DIFF), gsvc_fn(PixA, AllocatedL, KerStack) ->
    convolved image and bdc_fn(**kw) -> kernel default to the CPU restatements; the GPU test passes the HIP operators."""
    import time
    from . import gridconv_oracle as GO
    t0 = time.time()
    out = {}
    REF, SCI = cross_convolve(case, workers)
    mREF, mSCI = masked_pair(case, REF, SCI)
    N0, N1 = REF.shape
    s = notebook_settings(N0, N1)
    if bsp_fn is None:
        Solution, DIFF = bsp(REF, SCI, mREF, mSCI, s, workers)[:2]
    else:
        Solution, DIFF = bsp_fn(REF, SCI, mREF, mSCI, s)
    out.update(REF=REF, SCI=SCI, mREF=mREF, mSCI=mSCI, settings=s, Solution=Solution, DIFF=DIFF)
    if verbose:
        print("[nircam] subtraction done %.1f s" % (time.time() - t0))
    AllocatedL, XY_TiC = GO.tile_labels(N0, N1, round(TILESIZE_RATIO * GKERHW))
    MKerStack = matching_kernels(Solution, s, N0, N1, XY_TiC)
    DCKerStack, sig = decorrelation_kernels(case, MKerStack, bdc_fn)
    out.update(MKerStack=MKerStack, DCKerStack=DCKerStack, bkgsig=sig, AllocatedL=AllocatedL, XY_TiC=XY_TiC)
    if verbose:
        print("[nircam] kernels done %.1f s" % (time.time() - t0))
    PixA_in = np.array(DIFF, dtype=np.float64)
    PixA_in[np.isnan(PixA_in)] = 0.0
    if gsvc_fn is None:
        DCDIFF = GO.gsvc(PixA_in, AllocatedL, DCKerStack, normalize_kernel=True, use_fft=True)
    else:
        DCDIFF = gsvc_fn(PixA_in, AllocatedL, DCKerStack)
    B = boundary_mask(N0, N1)
    DCDIFF[B] = 0.
    out["DCDIFF"] = DCDIFF
    if verbose:
        print("[nircam] grid convolution done %.1f s" % (time.time() - t0))
    NoiseD = noise_map(case, MKerStack, DCKerStack, workers)
    out["NoiseD"] = NoiseD
    out["SNR"] = DCDIFF / NoiseD
    if verbose:
        print("[nircam] noise map done %.1f s" % (time.time() - t0))
    return out
