"""Helpers shared by the CPU and GPU parity tests: golden fixture loading and error metrics."""
import ast
import glob
import os

import numpy as np

from sfft_amd.utils.synthetic import make_pair, pair_checksum

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    """Fixtures of the polynomial path (sfft/sfftcore), made by make_golden.py."""
    return sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(GOLDEN_DIR, "c*.npz")))


def bspline_golden_names():
    """Fixtures of the B-spline path, made by make_golden_bspline.py."""
    return sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(GOLDEN_DIR, "bs_*.npz")))


def load_bspline_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files if k != "meta"}
    g["meta"] = ast.literal_eval(str(z["meta"][0]))
    return g


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = ast.literal_eval(str(z["meta"][0]))
    g = {k: z[k] for k in z.files if k != "meta"}
    g["meta"] = meta
    if "LHMAT_tril" in g:           # big systems are stored as the packed lower triangle (asymmetry of the original in meta)
        n = g["Solution"].shape[0]
        LH = np.zeros((n, n))
        LH[np.tril_indices(n)] = g.pop("LHMAT_tril")
        g["LHMAT"] = LH + np.tril(LH, -1).T
    if "REF" not in g:
        pair = make_pair(meta["N0"], meta["N1"], seed=meta["seed"], mask=bool(meta["mask"]),
                         nan_pixels=meta["nan_pixels"], sky=meta["sky"], bkg_scale=meta["bkg_scale"])
        cs = pair_checksum(pair)
        assert abs(cs - meta["checksum"]) <= 1e-9 * abs(meta["checksum"]), "seeded inputs did not regenerate"
        g.update(pair)
    return g


def packet_roles(g):
    """Apply the packet's NaN fill and ForceConv role swap (CustomizedPacket.py:114-162).
    Returns I, J, mI, mJ, NaNmask_U."""
    REF, SCI, mREF, mSCI = g["REF"], g["SCI"], g["mREF"], g["mSCI"]
    nm = None
    if np.isnan(REF).any() or np.isnan(SCI).any():
        nm = np.isnan(REF) | np.isnan(SCI)
    if g["meta"]["ForceConv"] == "REF":
        mI, mJ, I, J = mREF, mSCI, REF, SCI
    else:
        mI, mJ, I, J = mSCI, mREF, SCI, REF
    if nm is not None:
        I, J = I.copy(), J.copy()
        I[nm] = mI[nm]
        J[nm] = mJ[nm]
    return I, J, mI, mJ, nm


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.nanmean(a * a)))


def rel_rms_err(a, ref):
    return rms(np.asarray(a) - np.asarray(ref)) / max(rms(ref), 1e-300)
