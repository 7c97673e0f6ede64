#!/usr/bin/env python3
"""Golden vectors for the B-spline form of SFFT, from the REFERENCE's own Numpy implementation of it.

sfft/BSplineSFFT.py is CuPy-only.  The reference's importable CPU implementation of the same algorithm is the
development version under misc/beta4spline/new_version_sfftcore (B-spline or polynomial kernel / background
variation, ConstPhotRatio False = BSplineSFFT 'ENTANGLED' scaling, True = 'SEPARATE-CONSTANT'; no regularisation).
Run in the build container only:

    python tests/golden/make_golden_bspline.py

numba / pyfftw stand-ins as in make_golden.py; the reference code runs unmodified.  Fixtures tests/golden/bs_*.npz hold
inputs (REF/SCI/mREF/mSCI), the configuration (meta), LHMAT/RHb as handed to the reference's TweakLS, Solution, DIFF.

B-spline BACKGROUND variation: as shipped, the dev version's Numpy branch for it raises UnboundLocalError -- SFFTConfigure.py:1249
defines the background function under the name `KerSpatial` and line 1268 then registers the unbound name `BkgSpatial`.  The cases
marked `patch_bkg_name` run the SAME reference code with that one identifier corrected IN MEMORY at generation time (the source file
is read from /root/reference, the one `def` line is renamed, the module is exec'd; nothing of the source is stored here): every
function body is the reference's.  These fixtures (bs_*_bkgbspl*.npz) are therefore "reference code with a one-token fix", which is
what the tests' docstrings and DESIGN section 2 say about them.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFDIR = "/root/reference/misc/beta4spline/new_version_sfftcore"
sys.path.insert(0, ROOT)
from sfft_amd.utils.synthetic import make_pair  # noqa: E402


def load_reference(patch_bkg_name=False):
    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.njit = njit
    nb.prange = range
    sys.modules["numba"] = nb
    fw = types.ModuleType("pyfftw")
    fw.config = types.SimpleNamespace(NUM_THREADS=1)
    fw.interfaces = types.ModuleType("pyfftw.interfaces")
    fw.interfaces.cache = types.SimpleNamespace(enable=lambda: None)
    fw.interfaces.numpy_fft = np.fft
    sys.modules["pyfftw"] = fw
    sys.modules["pyfftw.interfaces"] = fw.interfaces
    mods = {}
    for name in ("SFFTConfigure", "SFFTSubtract"):
        path = os.path.join(REFDIR, name + ".py")
        if patch_bkg_name and name == "SFFTConfigure":
            # the one-token fix: the B-spline background function is defined under the wrong name (see the module docstring)
            src = open(path).read()
            bad = "def KerSpatial(REF_pq, BkgSplBasisX, BkgSplBasisY, SPixA_Tpq):"
            assert src.count(bad) == 1
            src = src.replace(bad, "def BkgSpatial(REF_pq, BkgSplBasisX, BkgSplBasisY, SPixA_Tpq):")
            m = types.ModuleType("refbs_patched_" + name)
            m.__file__ = path
            exec(compile(src, path, "exec"), m.__dict__)
        else:
            spec = importlib.util.spec_from_file_location("refbs_" + name, path)
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
        mods[name] = m
    return mods["SFFTConfigure"], mods["SFFTSubtract"]


CASES = [
    dict(name="bs_48x40_w2_bspl2_k1_poly1_const", N0=48, N1=40, w=2, KerSpType="B-Spline", KerSpDegree=2,
         KerIntKnotX=[24.5], KerIntKnotY=[20.5], BkgSpType="Polynomial", BkgSpDegree=1, BkgIntKnotX=[], BkgIntKnotY=[],
         CPR=True, seed=31, mask=True),
    dict(name="bs_48x40_w2_bspl2_k1_poly1_entangled", N0=48, N1=40, w=2, KerSpType="B-Spline", KerSpDegree=2,
         KerIntKnotX=[24.5], KerIntKnotY=[20.5], BkgSpType="Polynomial", BkgSpDegree=1, BkgIntKnotX=[], BkgIntKnotY=[],
         CPR=False, seed=32, mask=True),
    dict(name="bs_64x48_w2_bspl1_k2_poly2_const", N0=64, N1=48, w=2, KerSpType="B-Spline", KerSpDegree=1,
         KerIntKnotX=[20.5, 44.5], KerIntKnotY=[24.5], BkgSpType="Polynomial", BkgSpDegree=2, BkgIntKnotX=[], BkgIntKnotY=[],
         CPR=True, seed=33, mask=False),
    dict(name="bs_45x35_w1_bspl3_k0_poly0_const", N0=45, N1=35, w=1, KerSpType="B-Spline", KerSpDegree=3,
         KerIntKnotX=[], KerIntKnotY=[], BkgSpType="Polynomial", BkgSpDegree=0, BkgIntKnotX=[], BkgIntKnotY=[],
         CPR=True, seed=34, mask=True),
    # B-spline BACKGROUND variation (reference code with the one-token name fix, see the module docstring)
    dict(name="bs_64x48_w2_bspl2_k1_bkgbspl2_k1_const", N0=64, N1=48, w=2, KerSpType="B-Spline", KerSpDegree=2,
         KerIntKnotX=[32.5], KerIntKnotY=[24.5], BkgSpType="B-Spline", BkgSpDegree=2, BkgIntKnotX=[30.5], BkgIntKnotY=[22.5],
         CPR=True, seed=36, mask=True, patch_bkg_name=True),
    dict(name="bs_60x56_w2_poly1_bkgbspl1_k2_entangled", N0=60, N1=56, w=2, KerSpType="Polynomial", KerSpDegree=1,
         KerIntKnotX=[], KerIntKnotY=[], BkgSpType="B-Spline", BkgSpDegree=1, BkgIntKnotX=[20.5, 40.5], BkgIntKnotY=[28.5],
         CPR=False, seed=37, mask=False, patch_bkg_name=True),
    dict(name="bs_48x64_w3_bspl1_k1_bkgbspl3_k0_const", N0=48, N1=64, w=3, KerSpType="B-Spline", KerSpDegree=1,
         KerIntKnotX=[24.5], KerIntKnotY=[32.5], BkgSpType="B-Spline", BkgSpDegree=3, BkgIntKnotX=[], BkgIntKnotY=[],
         CPR=True, seed=38, mask=True, patch_bkg_name=True),
    # polynomial kernel through the same code path (TweakLS = deletion)
    dict(name="bs_64x64_w2_poly2_poly1_const", N0=64, N1=64, w=2, KerSpType="Polynomial", KerSpDegree=2,
         KerIntKnotX=[], KerIntKnotY=[], BkgSpType="Polynomial", BkgSpDegree=1, BkgIntKnotX=[], BkgIntKnotY=[],
         CPR=True, seed=35, mask=True),
]


def run_case(cfgmod, submod, c):
    pair = make_pair(c["N0"], c["N1"], seed=c["seed"], mask=c["mask"], sky=0.0 if c["mask"] else 100.0,
                     bkg_scale=0.05 if c["mask"] else 1.0, density=400.0)
    cfg = cfgmod.SingleSFFTConfigure.SSC(NX=c["N0"], NY=c["N1"], KerHW=c["w"], KerSpType=c["KerSpType"],
                                         KerSpDegree=c["KerSpDegree"], KerIntKnotX=c["KerIntKnotX"], KerIntKnotY=c["KerIntKnotY"],
                                         BkgSpType=c["BkgSpType"], BkgSpDegree=c["BkgSpDegree"], BkgIntKnotX=c["BkgIntKnotX"],
                                         BkgIntKnotY=c["BkgIntKnotY"], ConstPhotRatio=c["CPR"], BACKEND_4SUBTRACT="Numpy",
                                         NUM_CPU_THREADS_4SUBTRACT=1, VERBOSE_LEVEL=0)
    pdict, mdict = cfg
    cap = {}
    orig_del, orig_phi = mdict["FillLS_DEL"], mdict["FillLS_PHI"]

    def cap_del(PreDEL, RHb):
        out = orig_del(PreDEL=PreDEL, RHb=RHb)
        cap["RHb"] = np.array(out, copy=True)
        return out

    def cap_phi(PrePHI, LHMAT):
        out = orig_phi(PrePHI=PrePHI, LHMAT=LHMAT)
        cap["LHMAT"] = np.array(out, copy=True)
        return out
    mdict["FillLS_DEL"], mdict["FillLS_PHI"] = cap_del, cap_phi
    Solution, DIFF, _ = submod.GeneralSFFTSubtract.GSS(
        PixA_I=pair["REF"], PixA_J=pair["SCI"], PixA_mI=pair["mREF"], PixA_mJ=pair["mSCI"], SFFTConfig=cfg,
        ContamMask_I=None, BACKEND_4SUBTRACT="Numpy", NUM_CPU_THREADS_4SUBTRACT=1, VERBOSE_LEVEL=0)
    meta = {k: v for k, v in c.items() if k != "name"}
    meta.update(NEQ=int(pdict["NEQ"]), Fij=int(pdict["Fij"]), Fpq=int(pdict["Fpq"]))
    np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), meta=np.array([repr(meta)]), REF=pair["REF"], SCI=pair["SCI"],
                        mREF=pair["mREF"], mSCI=pair["mSCI"], LHMAT=cap["LHMAT"], RHb=cap["RHb"], Solution=Solution, DIFF=DIFF)
    print("%-44s NEQ=%4d Fij=%2d Fpq=%2d cond=%.2e rms(DIFF)=%.4g" % (c["name"], pdict["NEQ"], pdict["Fij"], pdict["Fpq"],
          np.linalg.cond(cap["LHMAT"]), np.sqrt(np.mean(DIFF ** 2))), flush=True)


if __name__ == "__main__":
    only = sys.argv[1:]
    for patched in (False, True):
        cfgmod, submod = load_reference(patch_bkg_name=patched)
        for c in CASES:
            if bool(c.get("patch_bkg_name")) == patched and (not only or c["name"] in only):
                run_case(cfgmod, submod, c)
