"""Seeded synthetic reference/science image pairs for parity tests and bench.py.

The recipe follows SURVEY.md section 8(d): a star field of circular Gaussians on a
flat sky with white noise (REF), and a science frame that is the noiseless
reference blurred by a small Gaussian, scaled by a photometric ratio, plus a
low-order polynomial differential background and independent noise (SCI).
Stars are rendered analytically (a Gaussian convolved with a Gaussian is a
Gaussian), with circular wrap at the image edges so that the pair obeys the
periodic model the SFFT core assumes.

Axis 0 is the FITS X axis (rows of the C-ordered array), as in the reference
packets, which transpose FITS data on read (sfft/CustomizedPacket.py:93).
"""
import numpy as np

__all__ = ["make_pair", "pair_checksum"]


def _render(N0, N1, x0, x1, flux, sigma, half=None):
    """Add wrapped circular Gaussians of total flux `flux` at (x0, x1)."""
    img = np.zeros((N0, N1), dtype=np.float64)
    if half is None:
        half = int(np.ceil(5.0 * sigma)) + 1
    off = np.arange(-half, half + 1)
    norm = 1.0 / (2.0 * np.pi * sigma * sigma)
    for cx, cy, f in zip(x0, x1, flux):
        ix, iy = int(np.floor(cx)), int(np.floor(cy))
        rx = (ix + off)
        ry = (iy + off)
        gx = np.exp(-0.5 * ((rx - cx) / sigma) ** 2)
        gy = np.exp(-0.5 * ((ry - cy) / sigma) ** 2)
        stamp = (f * norm) * np.outer(gx, gy)
        np.add.at(img, (np.mod(rx, N0)[:, None], np.mod(ry, N1)[None, :]), stamp)
    return img


def make_pair(N0, N1, seed=1234, mask=True, nan_pixels=0, density=2500.0,
              phot_ratio=1.3, sigma_ref=1.1, sigma_match=0.9, noise=3.0, sky=100.0, bkg_scale=1.0):
    """Return dict with REF, SCI, mREF, mSCI (float64, C order, shape (N0, N1)).

    mask=True  -> masked pair has pixels far from every star zeroed in both frames
    mask=False -> masked pair is identical to the full pair
    nan_pixels -> that many NaNs are put in REF and SCI (never in the masked pair)
    sky, bkg_scale -> flat sky under REF and amplitude of the differential background in SCI
    """
    rng = np.random.default_rng(seed)
    P = N0 * N1
    nstar = max(4, int(round(P / density)))
    x0 = rng.uniform(0, N0, nstar)
    x1 = rng.uniform(0, N1, nstar)
    flux = 10.0 ** rng.uniform(2.5, 4.5, nstar)

    ref_clean = _render(N0, N1, x0, x1, flux, sigma_ref)
    sigma_sci = float(np.hypot(sigma_ref, sigma_match))
    sci_clean = _render(N0, N1, x0, x1, phot_ratio * flux, sigma_sci)

    cx = (np.arange(N0, dtype=np.float64)[:, None] + 1.0) / N0
    cy = (np.arange(N1, dtype=np.float64)[None, :] + 1.0) / N1
    bkg = bkg_scale * (20.0 + 5.0 * cx - 3.0 * cy * cy)

    REF = ref_clean + sky + rng.normal(0.0, noise, (N0, N1))
    SCI = sci_clean + phot_ratio * sky + bkg + rng.normal(0.0, noise, (N0, N1))

    if mask:
        # keep pixels within a flux-dependent radius of any star
        keep = np.zeros((N0, N1), dtype=bool)
        rad = np.clip(np.round(3.0 * np.log10(flux)).astype(int), 3, 14)
        for cx_, cy_, r in zip(x0, x1, rad):
            ix, iy = int(np.floor(cx_)), int(np.floor(cy_))
            o = np.arange(-r, r + 1)
            disk = (o[:, None] ** 2 + o[None, :] ** 2) <= r * r
            rows = np.mod(ix + o, N0)[:, None]
            cols = np.mod(iy + o, N1)[None, :]
            keep[rows, cols] |= disk
        mREF = np.where(keep, REF, 0.0)
        mSCI = np.where(keep, SCI, 0.0)
    else:
        mREF = REF.copy()
        mSCI = SCI.copy()

    if nan_pixels:
        idx = rng.choice(P, size=2 * nan_pixels, replace=False)
        REF = REF.copy()
        SCI = SCI.copy()
        REF.reshape(-1)[idx[:nan_pixels]] = np.nan
        SCI.reshape(-1)[idx[nan_pixels:]] = np.nan

    out = dict(REF=REF, SCI=SCI, mREF=mREF, mSCI=mSCI)
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in out.items()}


def pair_checksum(pair):
    """Order-sensitive fp64 checksum used to verify that seeded inputs regenerate identically."""
    acc = 0.0
    for k in ("REF", "SCI", "mREF", "mSCI"):
        a = np.nan_to_num(pair[k], nan=-7.0).reshape(-1)
        w = np.cos(np.arange(a.size, dtype=np.float64) * 0.001)
        acc += float(np.dot(a, w))
    return acc
