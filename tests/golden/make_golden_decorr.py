#!/usr/bin/env python3
"""Golden vectors for the noise-decorrelation kernel from the reference's DeCorrelation_Calculator.DCC,
run on the reference's own test inputs (test/difference_noise_decorrelation/input_data) and checked against its own
golden output (4check/DeCorrKernel.fits).  Build container only.

The reference modules are loaded by path; `sfft` / `sfft.utils` are registered as empty packages first so that
`from sfft.utils.ConvKernelConvertion import ...` resolves without importing sfft/__init__ (astropy, ...).
FITS files are read with this repo's minimal FITS reader (astropy is absent).  Sky sigmas come from the reference's
SkyLevel_Estimator.SLE exactly as its decorr.py script computes them.
"""
import importlib.util, os, sys, types, io, contextlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from sfft_amd.utils import minifits  # noqa: E402
REF = "/root/reference"
for name in ("sfft", "sfft.utils"):
    m = types.ModuleType(name); m.__path__ = []; sys.modules[name] = m


def load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec); sys.modules[modname] = m; spec.loader.exec_module(m); return m


CKC = load("sfft.utils.ConvKernelConvertion", "sfft/utils/ConvKernelConvertion.py")
DCCm = load("sfft.utils.DeCorrelationCalculator", "sfft/utils/DeCorrelationCalculator.py")
SLE = load("sfft.utils.SkyLevelEstimator", "sfft/utils/SkyLevelEstimator.py")
D = os.path.join(REF, "test/difference_noise_decorrelation")
rd = lambda f: np.ascontiguousarray(minifits.getdata(os.path.join(D, f))[0].T, dtype=np.float64)
sci = ["DEC-OBS04%s" % c for c in "abcde"]
ref = ["DEC-OBS18%s" % c for c in "abcde"]
out = {}
sig_S, mk_S, sig_R, mk_R = [], [], [], []
with contextlib.redirect_stdout(io.StringIO()):
    for grp, sigs, mks in ((sci, sig_S, mk_S), (ref, sig_R, mk_R)):
        for k, base in enumerate(grp):
            sigs.append(float(SLE.SkyLevel_Estimator.SLE(PixA_obj=rd("input_data/%s.mini.fits" % base))[1]))
            mks.append(None if k == 0 else rd("input_data/%s.MatchKernel.fits" % base))
    mk_fin = rd("input_data/FinalMatchKernel.fits")
    K = DCCm.DeCorrelation_Calculator.DCC(MK_JLst=mk_S, SkySig_JLst=sig_S, MK_ILst=mk_R, SkySig_ILst=sig_R, MK_Fin=mk_fin,
                                          KERatio=2.0, VERBOSE_LEVEL=0)
    K_stack = DCCm.DeCorrelation_Calculator.DCC(MK_JLst=mk_S, SkySig_JLst=sig_S, KERatio=1.5, VERBOSE_LEVEL=0)
check = rd("4check/DeCorrKernel.fits")
print("reference DCC vs its own 4check golden: max abs diff %.3e (max |K| %.3e)" % (np.abs(K - check).max(), np.abs(check).max()))
assert np.abs(K - check).max() <= 1e-5 * np.abs(check).max()      # 4check was written by an older sfft (float32 file; sky sigmas differ in the last digits)
for k in range(5):
    out["sigS%d" % k] = sig_S[k]; out["sigR%d" % k] = sig_R[k]
    if k > 0:
        out["mkS%d" % k] = mk_S[k]; out["mkR%d" % k] = mk_R[k]
out["mkFin"] = mk_fin; out["KDeCo_sub"] = K; out["KDeCo_stack"] = K_stack
np.savez_compressed(os.path.join(HERE, "decorr_case.npz"), **out)
print("wrote decorr_case.npz: kernels %s, KDeCo %s / %s" % (mk_fin.shape, K.shape, K_stack.shape))
