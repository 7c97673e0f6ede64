#!/usr/bin/env python3
"""Golden vectors for the Solution decoding from the reference's sfft/utils/SFFTSolutionReader.py.
Build container only.  The module imports astropy.io.fits at the top (absent here, used only by its FITS variants), so a
stand-in `astropy.io.fits` module is registered before the import; the array code runs unmodified."""
import importlib.util, os, sys, types
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ap = types.ModuleType("astropy"); io = types.ModuleType("astropy.io"); fits = types.ModuleType("astropy.io.fits")
ap.io = io; io.fits = fits
sys.modules.update({"astropy": ap, "astropy.io": io, "astropy.io.fits": fits})
spec = importlib.util.spec_from_file_location("ref_reader", "/root/reference/sfft/utils/SFFTSolutionReader.py")
R = importlib.util.module_from_spec(spec); spec.loader.exec_module(R)
rng = np.random.default_rng(41)
out = {}
for k, (N0, N1, w, DK, DB) in enumerate([(128, 96, 2, 2, 2), (4096, 4096, 8, 2, 2), (64, 64, 1, 0, 0), (300, 200, 3, 3, 1)]):
    L = 2 * w + 1
    Fij, Fpq = (DK + 1) * (DK + 2) // 2, (DB + 1) * (DB + 2) // 2
    sol = rng.normal(size=Fij * L * L + Fpq) * N0 * N1 * 0.01
    XY = np.stack([rng.uniform(0.5, N0 + 0.5, 7), rng.uniform(0.5, N1 + 0.5, 7)], axis=1)
    d = R.Read_SFFTSolution().FromArray(sol, N0, N1, L, L, DK, Fpq)
    st = R.SVKDict_SFFT2ST.convert(DK, DK, d)
    back = R.SVKDict_ST2SFFT.convert(DK, DK, st)
    out["case%d_meta" % k] = np.array([N0, N1, w, DK, DB])
    out["case%d_sol" % k] = sol; out["case%d_xy" % k] = XY
    out["case%d_sfft" % k] = np.array([d[ij] for ij in sorted(d)])
    out["case%d_std" % k] = np.array([st[ij] for ij in sorted(st)])
    out["case%d_back" % k] = np.array([back[ij] for ij in sorted(back)])
    out["case%d_kers" % k] = R.Realize_MatchingKernel(XY).FromArray(sol, N0, N1, L, L, DK, Fpq)
    out["case%d_fscal" % k] = R.Realize_FluxScaling(XY).FromArray(sol, N0, N1, L, L, DK, Fpq)
np.savez_compressed(os.path.join(HERE, "reader_cases.npz"), **out)
print("wrote reader_cases.npz", len(out))
