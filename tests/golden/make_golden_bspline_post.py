#!/usr/bin/env python3
"""Golden vectors for the numpy-only post-processing classes of the reference's sfft/BSplineSFFT.py:
Read_SFFTSolution.FromArray (:4417-4523), BSpline_MatchingKernel.FromArray (:4561-4662), BSpline_DeCorrelation.BDC (:4755-4868).
Build container only.  The module's top-level imports that are absent here (astropy.io.fits, astropy.convolution,
sfft.utils.meta.MultiProc -- none of them used by the three functions above) get empty stand-in modules; the reference
code itself runs unmodified."""
import importlib.util, os, sys, types
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


stub("astropy"); stub("astropy.io"); stub("astropy.io.fits")
stub("astropy.convolution", convolve=None, convolve_fft=None)
stub("sfft"); stub("sfft.utils"); stub("sfft.utils.meta"); stub("sfft.utils.meta.MultiProc", Multi_Proc=None)
spec = importlib.util.spec_from_file_location("ref_bspl", "/root/reference/sfft/BSplineSFFT.py")
R = importlib.util.module_from_spec(spec); spec.loader.exec_module(R)

rng = np.random.default_rng(77)
out = {}
CASES = [
    # N0, N1, w, KerSpType, DK, KnotX, KnotY, SEPARATE_SCALING, ScaSpType, DS, ScaKnotX, ScaKnotY, Fpq
    (128, 96, 2, 'Polynomial', 2, [], [], False, None, None, [], [], 6),
    (128, 96, 2, 'Polynomial', 2, [], [], True, 'Polynomial', 0, [], [], 3),
    (200, 160, 3, 'B-Spline', 2, [100.5], [60.5, 110.5], True, 'Polynomial', 0, [], [], 6),
    (200, 160, 2, 'B-Spline', 2, [100.5], [80.5], True, 'Polynomial', 1, [], [], 6),
    (256, 256, 3, 'B-Spline', 3, [128.5], [], True, 'B-Spline', 1, [128.5], [], 4),
    (96, 128, 1, 'Polynomial', 3, [], [], True, 'B-Spline', 2, [], [], 1),
    (64, 64, 2, 'B-Spline', 1, [32.5], [32.5], False, None, None, [], [], 3),
]
for k, (N0, N1, w, kt, DK, kx, ky, sep, st, DS, sx, sy, Fpq) in enumerate(CASES):
    L = 2 * w + 1
    if kt == 'Polynomial':
        Fi = Fj = -1
        Fij = (DK + 1) * (DK + 2) // 2
    else:
        Fi, Fj = len(kx) + DK + 1, len(ky) + DK + 1
        Fij = Fi * Fj
    ScaFi = ScaFj = None
    if sep and DS > 0:
        if st == 'Polynomial':
            ScaFi = ScaFj = -1
        else:
            ScaFi, ScaFj = len(sx) + DS + 1, len(sy) + DS + 1
    sol = rng.normal(size=Fij * L * L + Fpq) * N0 * N1 * 0.01
    XY = np.stack([rng.uniform(0.5, N0 + 0.5, 9), rng.uniform(0.5, N1 + 0.5, 9)], axis=1)
    kd, sd = R.Read_SFFTSolution().FromArray(Solution=sol, KerSpType=kt, N0=N0, N1=N1, DK=DK, L0=L, L1=L, Fi=Fi, Fj=Fj, Fpq=Fpq,
                                             SEPARATE_SCALING=sep, ScaSpType=st, DS=DS, ScaFi=ScaFi, ScaFj=ScaFj)
    ks = R.BSpline_MatchingKernel(XY_q=XY, VERBOSE_LEVEL=0).FromArray(
        Solution=sol, KerSpType=kt, KerIntKnotX=kx, KerIntKnotY=ky, N0=N0, N1=N1, DK=DK, L0=L, L1=L, Fi=Fi, Fj=Fj, Fpq=Fpq,
        SEPARATE_SCALING=sep, ScaSpType=st, ScaIntKnotX=sx, ScaIntKnotY=sy, DS=DS, ScaFi=ScaFi, ScaFj=ScaFj)
    out["c%d_meta" % k] = np.array([repr(dict(N0=N0, N1=N1, w=w, KerSpType=kt, DK=DK, KerIntKnotX=kx, KerIntKnotY=ky, SEPARATE_SCALING=sep,
                                              ScaSpType=st, DS=DS, ScaIntKnotX=sx, ScaIntKnotY=sy, Fpq=Fpq, Fi=Fi, Fj=Fj, ScaFi=ScaFi, ScaFj=ScaFj))])
    out["c%d_sol" % k] = sol
    out["c%d_xy" % k] = XY
    keys = list(kd.keys())
    out["c%d_kerkeys" % k] = np.array(keys)
    out["c%d_kerdict" % k] = np.array([kd[t] for t in keys])
    if sd is not None:
        skeys = list(sd.keys())
        out["c%d_scakeys" % k] = np.array(skeys)
        out["c%d_scadict" % k] = np.array([sd[t] for t in skeys])
    out["c%d_kerstack" % k] = ks

# noise decorrelation from realised kernels: image-subtraction and image-stacking modes
def gk(L, s, dx=0.0):
    a = np.arange(L) - (L - 1) / 2
    g = np.exp(-0.5 * ((a[:, None] - dx) ** 2 + a[None, :] ** 2) / s ** 2)
    return g / g.sum()
mkj, mki, mkf = gk(9, 1.3, 0.4), gk(7, 1.0), gk(11, 1.8, -0.3)
out["bdc_mkj"], out["bdc_mki"], out["bdc_mkf"] = mkj, mki, mkf
out["bdc_sub"] = R.BSpline_DeCorrelation.BDC(MK_JLst=[mkj], SkySig_JLst=[3.0], MK_ILst=[mki], SkySig_ILst=[2.0], MK_Fin=mkf,
                                             KERatio=2.0, DENO_CLIP_RATIO=100000.0, VERBOSE_LEVEL=0)
out["bdc_sub_nofin"] = R.BSpline_DeCorrelation.BDC(MK_JLst=[None], SkySig_JLst=[3.0], MK_ILst=[mki], SkySig_ILst=[2.0], MK_Fin=None,
                                                   KERatio=1.5, DENO_CLIP_RATIO=1000.0, VERBOSE_LEVEL=0)
out["bdc_stack"] = R.BSpline_DeCorrelation.BDC(MK_JLst=[mkj, None, mkf], SkySig_JLst=[3.0, 2.5, 4.0], KERatio=2.0, VERBOSE_LEVEL=0)
np.savez_compressed(os.path.join(HERE, "bspline_post_cases.npz"), **out)
print("wrote bspline_post_cases.npz", len(out), {k: v.shape for k, v in out.items() if k.startswith("bdc")})
