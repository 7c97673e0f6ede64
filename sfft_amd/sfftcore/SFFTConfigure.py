"""SingleSFFTConfigure.SSC -- mirror of sfft/sfftcore/SFFTConfigure.py:1369-1397 for the HIP backend.

The reference's "compile" step JITs 17 kernels with the image size baked in on every call.  Here it
returns the same `SFFTConfig = (SFFTParam_dict, SFFTModule_dict)` tuple; the parameter dictionary has
the reference's keys (SFFTConfigure.py:50-75, read back by callers at sfft/CustomizedPacket.py:207-217)
and the module dictionary carries the cached `sfft_plan` handle instead of kernel objects.
"""
import numpy as np

from ..plan import get_plan

# names accepted for the GPU backend: 'Cupy' is what reference callers pass for "the GPU path"
_GPU_BACKENDS = ("Cupy", "HIP")


def _param_dict(N0, N1, KerHW, DK, DB, ConstPhotRatio):
    """SFFTConfigure.py:34-75."""
    w0, w1 = int(KerHW), int(KerHW)
    L0, L1 = 2 * w0 + 1, 2 * w1 + 1
    Fab = L0 * L1
    Fij = int((DK + 1) * (DK + 2) / 2)
    Fpq = int((DB + 1) * (DB + 2) / 2)
    SCALE = np.float64(1 / (N0 * N1))
    SCALE_L = np.float64(1 / SCALE)
    d = {}
    d['N0'], d['N1'], d['w0'], d['w1'], d['DK'], d['DB'] = N0, N1, w0, w1, DK, DB
    d['ConstPhotRatio'] = ConstPhotRatio
    d['MaxThreadPerB'] = 8
    d['L0'], d['L1'], d['Fab'], d['Fij'], d['Fpq'] = L0, L1, Fab, Fij, Fpq
    d['SCALE'], d['SCALE_L'] = SCALE, SCALE_L
    d['NEQ'] = Fij * Fab + Fpq
    d['Fijab'] = Fij * Fab
    d['NEQ_FSfree'] = d['NEQ'] - (Fij - 1)
    d['FOMG'], d['FGAM'], d['FTHE'] = Fij ** 2, Fij * Fpq, Fij
    d['FPSI'], d['FPHI'], d['FDEL'] = Fpq * Fij, Fpq ** 2, Fpq
    return d


class SingleSFFTConfigure:
    @staticmethod
    def SSC(NX, NY, KerHW, KerPolyOrder=2, BGPolyOrder=2, ConstPhotRatio=True,
            BACKEND_4SUBTRACT='Cupy', NUM_CPU_THREADS_4SUBTRACT=8, NUMBA_CACHE=True, VERBOSE_LEVEL=2,
            CUDA_DEVICE_4SUBTRACT=None):
        """Same arguments as the reference (CUDA_DEVICE_4SUBTRACT is an extra, optional device index; the
        reference selects the device with cupy.cuda.Device(...).use() before calling SSC).
        NUM_CPU_THREADS_4SUBTRACT / NUMBA_CACHE are accepted and ignored (no CPU path here)."""
        N0, N1 = int(NX), int(NY)
        DK, DB = int(KerPolyOrder), int(BGPolyOrder)
        if BACKEND_4SUBTRACT not in _GPU_BACKENDS:
            raise Exception("MeLOn ERROR: sfft_amd only provides the GPU backend (BACKEND_4SUBTRACT='Cupy'); "
                            "there is no CPU path, got %r" % (BACKEND_4SUBTRACT,))
        if DK not in [0, 1, 2, 3]:
            raise Exception('MeLOn ERROR: Input KerPolyOrder should be 0/1/2/3!')
        if DB not in [0, 1, 2, 3]:
            raise Exception('MeLOn ERROR: Input BGPolyOrder should be 0/1/2/3!')
        if (N0 < 8) or (N1 < 8):
            raise Exception('MeLOn ERROR: Input Image has dramatically small size!')
        if VERBOSE_LEVEL in [1, 2]:
            print('\n --//--//--//--//-- TRIGGER SFFT COMPILATION --//--//--//--//-- ')
            print('\n ---//--- KerPolyOrder %d | BGPolyOrder %d | KerHW [%d] ---//--- ' % (DK, DB, int(KerHW)))
        if CUDA_DEVICE_4SUBTRACT is None:
            import torch
            device = torch.cuda.current_device()
        else:
            device = int(CUDA_DEVICE_4SUBTRACT)
        plan = get_plan(N0, N1, int(KerHW), DK, DB, bool(ConstPhotRatio), device)
        SFFTParam_dict = _param_dict(N0, N1, int(KerHW), DK, DB, ConstPhotRatio)
        SFFTModule_dict = {'plan': plan, 'backend': 'HIP'}
        if VERBOSE_LEVEL in [1, 2]:
            print('\n --//--//--//--//-- EXIT SFFT COMPILATION --//--//--//--//-- ')
        return (SFFTParam_dict, SFFTModule_dict)
