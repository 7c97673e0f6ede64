#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Numpy backend.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [--only NAME]

The reference's `sfft/sfftcore/SFFTConfigure.py` and `SFFTSubtract.py` import only numpy
at module level and pull `numba` / `pyfftw` lazily; neither is installed here, so two
stand-in modules are registered before the import: `numba.njit` becomes the identity
decorator (`prange = range`) and `pyfftw.interfaces.numpy_fft` is `numpy.fft`.  The
reference code itself runs unmodified (SSC(...,'Numpy') -> GSS), in pure Python.

Each fixture (`tests/golden/<name>.npz`) holds inputs and expected outputs only:
  meta (N0,N1,KerHW,DK,DB,CPR,seed,...), REF/SCI/mREF/mSCI (or a seed + checksum for the
  large case), LHMAT, RHb (as handed to the reference's stripe removal / solver),
  Solution, DIFF.  No reference source text is stored.
"""
import argparse
import importlib.util
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from sfft_amd.utils.synthetic import make_pair, pair_checksum  # noqa: E402


def load_reference():
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
        spec = importlib.util.spec_from_file_location(
            "ref_" + name, os.path.join(REF, "sfft", "sfftcore", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["SFFTConfigure"], mods["SFFTSubtract"]


def C(name, N0, N1, w, DK, DB, CPR=True, FC="REF", seed=0, mask=True, nan=0, store=True, sky=0.0, bkg=0.05,
      contam=False, tril=False):
    return dict(name=name, N0=N0, N1=N1, w=w, DK=DK, DB=DB, CPR=CPR, FC=FC, seed=seed, mask=mask,
                nan=nan, store=store, sky=sky, bkg=bkg, contam=contam, tril=tril)


# masked cases mimic sky-subtracted frames (sky 0, faint differential background); unmasked ones
# carry a 100-count sky and the full background polynomial.
CASES = [
    C("c48x40_w2_k0b0_cpr", 48, 40, 2, 0, 0, seed=11),
    C("c48x40_w2_k1b1_cpr", 48, 40, 2, 1, 1, seed=12),
    C("c48x40_w2_k2b2_free", 48, 40, 2, 2, 2, CPR=False, seed=13, mask=False, sky=100.0, bkg=1.0),
    C("c64x64_w2_k2b2_cpr", 64, 64, 2, 2, 2, seed=14),
    C("c64x32_w3_k2b0_cpr", 64, 32, 3, 2, 0, FC="SCI", seed=15, mask=False, sky=100.0, bkg=0.0),
    C("c96x80_w3_k2b2_cpr_nan", 96, 80, 3, 2, 2, FC="SCI", seed=16, nan=7),
    C("c128x64_w2_k3b3_cpr", 128, 64, 2, 3, 3, seed=17, mask=False, sky=100.0, bkg=1.0),
    C("c45x35_w2_k1b2_free", 45, 35, 2, 1, 2, CPR=False, seed=18, sky=100.0, bkg=1.0),
    C("c128x96_w4_k2b2_cpr", 128, 96, 4, 2, 2, seed=19),
    # BASELINE.json config 1: 512x512, KerHW 4, constant kernel, flat background
    C("c512x512_w4_k0b0_cpr", 512, 512, 4, 0, 0, seed=20, store=False),
    # BASELINE.json config 2's kernel geometry (KerHW 8, orders 2/2, NEQ 1740) on a small frame; LHMAT is stored as its
    # packed lower triangle (the asymmetry of the reference's matrix is recorded in meta).  Several minutes each in pure Python.
    C("c96x80_w8_k2b2_cpr", 96, 80, 8, 2, 2, seed=21, tril=True, contam=True),
    C("c96x80_w8_k2b2_cpr_sci", 96, 80, 8, 2, 2, FC="SCI", seed=22, tril=True),
    # contamination-mask propagation through GSS (SFFTSubtract.py:907-921)
    C("c64x64_w3_k2b2_cpr_contam", 64, 64, 3, 2, 2, seed=23, contam=True),
    C("c48x40_w2_k1b1_free_contam", 48, 40, 2, 1, 1, CPR=False, seed=24, mask=False, sky=100.0, bkg=1.0, contam=True),
]


def run_case(cfgmod, submod, case):
    name, N0, N1, w, DK, DB, CPR, FC = (case[k] for k in ("name", "N0", "N1", "w", "DK", "DB", "CPR", "FC"))
    seed, mask, nnan, store = case["seed"], case["mask"], case["nan"], case["store"]
    pair = make_pair(N0, N1, seed=seed, mask=mask, nan_pixels=nnan, sky=case["sky"], bkg_scale=case["bkg"])
    REFa, SCIa, mREF, mSCI = pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"]

    cfg = cfgmod.SingleSFFTConfigure.SSC(NX=N0, NY=N1, KerHW=w, KerPolyOrder=DK, BGPolyOrder=DB,
                                         ConstPhotRatio=CPR, BACKEND_4SUBTRACT="Numpy",
                                         NUM_CPU_THREADS_4SUBTRACT=1, NUMBA_CACHE=False, VERBOSE_LEVEL=0)
    pdict, mdict = cfg
    cap = {}
    # capture the linear system as the reference hands it on (no change to its arithmetic)
    orig_del = mdict["FillLS_DEL"]
    orig_phi = mdict["FillLS_PHI"]

    def cap_del(PreDEL, RHb):
        out = orig_del(PreDEL=PreDEL, RHb=RHb)
        cap["RHb"] = np.array(out, copy=True)
        return out

    def cap_phi(PrePHI, LHMAT):
        out = orig_phi(PrePHI=PrePHI, LHMAT=LHMAT)
        cap["LHMAT"] = np.array(out, copy=True)   # OMG, GAM, PSI, PHI all filled at this point
        return out
    mdict["FillLS_DEL"] = cap_del
    mdict["FillLS_PHI"] = cap_phi

    # packet-level handling exactly as Customized_Packet.CP does it on arrays (CustomizedPacket.py:114-188)
    NaNmask_U = None
    if np.isnan(REFa).any() or np.isnan(SCIa).any():
        NaNmask_U = np.logical_or(np.isnan(REFa), np.isnan(SCIa))
    if FC == "REF":
        mI, mJ, I, J = mREF, mSCI, REFa, SCIa
    else:
        mI, mJ, I, J = mSCI, mREF, SCIa, REFa
    if NaNmask_U is not None:
        I, J = I.copy(), J.copy()
        I[NaNmask_U] = mI[NaNmask_U]
        J[NaNmask_U] = mJ[NaNmask_U]

    # contamination mask of image I (saturated cores and a bad column), propagated by GSS itself; the third ESS call of
    # GSS returns the convolved mask the threshold is applied to: it is captured (not altered) so that the test can
    # leave out pixels that sit on the threshold
    ContamMask_I = None
    ess_out = []
    if case["contam"]:
        ContamMask_I = I > np.percentile(I, 99.0)
        ContamMask_I[:, N1 // 3] = True
        ContamMask_I[N0 // 2, N1 // 2:N1 // 2 + 5] = True
        orig_ess = submod.ElementalSFFTSubtract.ESS

        def cap_ess(*a, **k):
            out = orig_ess(*a, **k)
            ess_out.append(out)
            return out
        submod.ElementalSFFTSubtract.ESS = staticmethod(cap_ess)

    t0 = time.time()
    try:
        Solution, DIFF, ContamMask_CI = submod.GeneralSFFTSubtract.GSS(
            PixA_I=I, PixA_J=J, PixA_mI=mI, PixA_mJ=mJ, SFFTConfig=cfg, ContamMask_I=ContamMask_I,
            BACKEND_4SUBTRACT="Numpy", NUM_CPU_THREADS_4SUBTRACT=1, VERBOSE_LEVEL=0)
    finally:
        if case["contam"]:
            submod.ElementalSFFTSubtract.ESS = staticmethod(orig_ess)
    if NaNmask_U is not None:
        DIFF[NaNmask_U] = np.nan
    if FC == "SCI":
        DIFF = -DIFF
    dt = time.time() - t0

    meta = dict(N0=N0, N1=N1, KerHW=w, DK=DK, DB=DB, CPR=int(CPR), seed=seed, mask=int(mask),
                nan_pixels=nnan, ForceConv=FC, sky=case["sky"], bkg_scale=case["bkg"],
                checksum=pair_checksum(pair))
    out = dict(meta=np.array([repr(meta)]), Solution=Solution, DIFF=DIFF,
               RHb=cap["RHb"])
    LH = cap["LHMAT"]
    if case["contam"]:
        assert len(ess_out) == 3 and ContamMask_CI is not None
        out.update(ContamMask_I=ContamMask_I, ContamMask_CI=ContamMask_CI, ContamD=np.array(ess_out[2][1], copy=True))
    if case["tril"]:
        meta["LHMAT_asym"] = float(np.max(np.abs(LH - LH.T)) / np.max(np.abs(LH)))
        out["meta"] = np.array([repr(meta)])
        out["LHMAT_tril"] = LH[np.tril_indices(LH.shape[0])]
    else:
        out["LHMAT"] = LH
    if store:
        out.update(REF=REFa, SCI=SCIa, mREF=mREF, mSCI=mSCI)
    # else: large case, inputs regenerate from the seed (checksum in meta)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    cond = np.linalg.cond(cap["LHMAT"])
    print("%-28s NEQ=%4d  cond=%.2e  rms(DIFF)=%.4g  %.1fs" %
          (name, pdict["NEQ"], cond, np.sqrt(np.nanmean(DIFF ** 2)), dt), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    cfgmod, submod = load_reference()
    for case in CASES:
        if args.only and case["name"] != args.only:
            continue
        run_case(cfgmod, submod, case)


if __name__ == "__main__":
    main()
