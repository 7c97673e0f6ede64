"""Customized_Packet.CP -- mirror of sfft/CustomizedPacket.py:12-223 on the HIP backend.

FITS paths in, (Solution, PixA_DIFF) numpy arrays out, optional FITS outputs with the reference's header
keywords.  File I/O uses astropy when importable and sfft_amd.utils.minifits otherwise.
"""
import os.path as pa
import time

import numpy as np
import torch

from ._packet_common import union_nan_mask, assign_roles, finish_diff
from .sfftcore.SFFTConfigure import SingleSFFTConfigure
from .sfftcore.SFFTSubtract import GeneralSFFTSubtract
from .utils import minifits

__all__ = ["Customized_Packet"]

try:  # pragma: no cover - astropy is absent from the target image
    from astropy.io import fits as _afits
except Exception:
    _afits = None


def _read_T(path):
    """fits.getdata(path, ext=0).T as float64 C-contiguous (CustomizedPacket.py:93-112)."""
    if _afits is not None:
        a = _afits.getdata(path, ext=0).T
    else:
        a = minifits.getdata(path)[0].T
    return np.ascontiguousarray(a, dtype=np.float64)


class Customized_Packet:
    @staticmethod
    def CP(FITS_REF, FITS_SCI, FITS_mREF, FITS_mSCI, ForceConv, GKerHW,
           FITS_DIFF=None, FITS_Solution=None, KerPolyOrder=2, BGPolyOrder=2, ConstPhotRatio=True,
           BACKEND_4SUBTRACT='Cupy', CUDA_DEVICE_4SUBTRACT='0', NUM_CPU_THREADS_4SUBTRACT=8, NUMBA_CACHE=True,
           VERBOSE_LEVEL=2):
        """Same parameters as the reference.  ForceConv='REF' -> DIFF = SCI - Conv(REF);
        ForceConv='SCI' -> DIFF = Conv(SCI) - REF."""
        PixA_REF = _read_T(FITS_REF)
        PixA_SCI = _read_T(FITS_SCI)
        PixA_mREF = _read_T(FITS_mREF)
        PixA_mSCI = _read_T(FITS_mSCI)

        NaNmask_U = union_nan_mask(np, PixA_REF, PixA_SCI)
        assert np.sum(np.isnan(PixA_mREF)) == 0
        assert np.sum(np.isnan(PixA_mSCI)) == 0
        ConvdSide, KerHW = ForceConv, GKerHW

        if BACKEND_4SUBTRACT not in ('Cupy', 'HIP'):
            raise Exception("MeLOn ERROR: sfft_amd only provides the GPU backend (BACKEND_4SUBTRACT='Cupy')")
        torch.cuda.set_device(int(CUDA_DEVICE_4SUBTRACT))

        if VERBOSE_LEVEL in [0, 1, 2]:
            print('MeLOn CheckPoint: TRIGGER Function Compilations of SFFT-SUBTRACTION!')
        Tcomp_start = time.time()
        SFFTConfig = SingleSFFTConfigure.SSC(NX=PixA_REF.shape[0], NY=PixA_REF.shape[1], KerHW=KerHW,
                                             KerPolyOrder=KerPolyOrder, BGPolyOrder=BGPolyOrder,
                                             ConstPhotRatio=ConstPhotRatio, BACKEND_4SUBTRACT=BACKEND_4SUBTRACT,
                                             NUM_CPU_THREADS_4SUBTRACT=NUM_CPU_THREADS_4SUBTRACT,
                                             NUMBA_CACHE=NUMBA_CACHE, VERBOSE_LEVEL=VERBOSE_LEVEL,
                                             CUDA_DEVICE_4SUBTRACT=int(CUDA_DEVICE_4SUBTRACT))
        if VERBOSE_LEVEL in [1, 2]:
            print('\nMeLOn Report: Function Compilations of SFFT-SUBTRACTION TAKES [%.3f s]' % (time.time() - Tcomp_start))

        PixA_I, PixA_J, PixA_mI, PixA_mJ = assign_roles(np, PixA_REF, PixA_SCI, PixA_mREF, PixA_mSCI, ConvdSide, NaNmask_U)

        if VERBOSE_LEVEL in [0, 1, 2]:
            print('MeLOn CheckPoint: TRIGGER SFFT-SUBTRACTION!')
        Tsub_start = time.time()
        _tmp = GeneralSFFTSubtract.GSS(PixA_I=PixA_I, PixA_J=PixA_J, PixA_mI=PixA_mI, PixA_mJ=PixA_mJ,
                                       SFFTConfig=SFFTConfig, ContamMask_I=None, BACKEND_4SUBTRACT=BACKEND_4SUBTRACT,
                                       NUM_CPU_THREADS_4SUBTRACT=NUM_CPU_THREADS_4SUBTRACT, VERBOSE_LEVEL=VERBOSE_LEVEL)
        Solution, PixA_DIFF = _tmp[:2]
        if VERBOSE_LEVEL in [1, 2]:
            print('\nMeLOn Report: SFFT-SUBTRACTION TAKES [%.3f s]' % (time.time() - Tsub_start))

        PixA_DIFF = finish_diff(PixA_DIFF, ConvdSide, NaNmask_U)

        # * Save difference image (CustomizedPacket.py:191-203): SCI's header + the SFFT keywords
        if FITS_DIFF is not None:
            kw = [('NAME_REF', pa.basename(FITS_REF)), ('NAME_SCI', pa.basename(FITS_SCI)),
                  ('KERORDER', KerPolyOrder), ('BGORDER', BGPolyOrder), ('CPHOTR', str(ConstPhotRatio)),
                  ('KERHW', KerHW), ('CONVD', ConvdSide)]
            if _afits is not None:
                _hdl = _afits.open(FITS_SCI)
                _hdl[0].data[:, :] = PixA_DIFF.T
                for k, v in kw:
                    _hdl[0].header[k] = (v, 'MeLOn: SFFT')
                _hdl.writeto(FITS_DIFF, overwrite=True)
                _hdl.close()
            else:
                data0, cards = minifits.getdata(FITS_SCI)
                for k, v in kw:
                    minifits.set_card(cards, k, v, 'MeLOn: SFFT')
                out = PixA_DIFF.T
                if data0.dtype.kind == 'f':
                    out = out.astype(data0.dtype.newbyteorder('='))
                minifits.writeto(FITS_DIFF, np.ascontiguousarray(out), cards)

        # * Save solution array (CustomizedPacket.py:205-221)
        if FITS_Solution is not None:
            P = SFFTConfig[0]
            kw = [('N0', P['N0']), ('N1', P['N1']), ('DK', P['DK']), ('DB', P['DB']), ('L0', P['L0']), ('L1', P['L1']),
                  ('FIJ', P['Fij']), ('FAB', P['Fab']), ('FPQ', P['Fpq']), ('FIJAB', P['Fijab'])]
            PixA_Solution = Solution.reshape((-1, 1))
            if _afits is not None:
                phdu = _afits.PrimaryHDU()
                for k, v in kw:
                    phdu.header[k] = (v, 'MeLOn: SFFT')
                phdu.data = PixA_Solution.T
                _afits.HDUList([phdu]).writeto(FITS_Solution, overwrite=True)
            else:
                cards = []
                for k, v in kw:
                    minifits.set_card(cards, k, int(v), 'MeLOn: SFFT')
                minifits.writeto(FITS_Solution, np.ascontiguousarray(PixA_Solution.T), cards)

        return Solution, PixA_DIFF
