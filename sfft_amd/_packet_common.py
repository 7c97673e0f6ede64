"""Array-level body shared by the three packets (Customized_Packet.CP, PureCupy_Customized_Packet.PCCP, BSpline_Packet.BSP).

Semantics follow the reference packets (sfft/CustomizedPacket.py:114-188, sfft/PureCupyCustomizedPacket.py:105-185):
  * the masked pair must be NaN-free; NaNs of REF/SCI form a union mask, are filled from the masked images for the
    subtraction and put back as NaN on the difference image;
  * ForceConv picks the convolved side: 'REF' -> DIFF = SCI - Conv(REF); 'SCI' -> DIFF = Conv(SCI) - REF, so transients
    on the science image are positive either way.
Works on numpy arrays and on torch tensors (`xp` = numpy or torch).
"""
import numpy as np


def union_nan_mask(xp, REF, SCI):
    nr, ns = xp.isnan(REF), xp.isnan(SCI)
    if bool(nr.any()) or bool(ns.any()):
        return xp.logical_or(nr, ns)
    return None


def assign_roles(xp, REF, SCI, mREF, mSCI, ForceConv, nanmask):
    """Return (I, J, mI, mJ): I is the image that gets convolved."""
    assert ForceConv in ['REF', 'SCI']
    if ForceConv == 'REF':
        I, J, mI, mJ = REF, SCI, mREF, mSCI
    else:
        I, J, mI, mJ = SCI, REF, mSCI, mREF
    if nanmask is not None:
        I = I.copy() if xp is np else I.clone()
        J = J.copy() if xp is np else J.clone()
        I[nanmask] = mI[nanmask]
        J[nanmask] = mJ[nanmask]
    return I, J, mI, mJ


def finish_diff(DIFF, ForceConv, nanmask):
    if nanmask is not None:
        DIFF[nanmask] = np.nan
    if ForceConv == 'SCI':
        DIFF = -DIFF
    return DIFF
