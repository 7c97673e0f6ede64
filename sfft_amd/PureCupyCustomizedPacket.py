"""PureCupy_Customized_Packet.PCCP -- mirror of sfft/PureCupyCustomizedPacket.py:38-187 on the HIP backend.

Device arrays in, device arrays out, no file I/O.  "GPU arrays" are torch.float64 CUDA (HIP) tensors (or
anything exposing __cuda_array_interface__); results are torch tensors on the same device.
"""
import time

import numpy as np
import torch

from ._packet_common import union_nan_mask, assign_roles, finish_diff
from .sfftcore.SFFTConfigure import SingleSFFTConfigure
from .sfftcore.SFFTSubtract import GeneralSFFTSubtract_PureCupy

__all__ = ["PureCupy_Customized_Packet"]


def _t(x, dev):
    """Everything the packet computes (NaN mask, role swap, DIFF) lives on CUDA_DEVICE_4SUBTRACT: inputs on another
    device, or on the host, are moved there first."""
    if isinstance(x, torch.Tensor):
        return x if x.device == dev else x.to(dev)
    return torch.as_tensor(x, device=dev)


class PureCupy_Customized_Packet:
    @staticmethod
    def PCCP(PixA_REF_GPU, PixA_SCI_GPU, PixA_mREF_GPU, PixA_mSCI_GPU, ForceConv, GKerHW,
             KerPolyOrder=2, BGPolyOrder=2, ConstPhotRatio=True, CUDA_DEVICE_4SUBTRACT='0', VERBOSE_LEVEL=2):
        """Same parameters and conventions as the reference:
        ForceConv='REF' -> DIFF = SCI - Conv(REF);  ForceConv='SCI' -> DIFF = Conv(SCI) - REF
        (transients on the science image are always positive on DIFF)."""
        dev = torch.device('cuda', int(CUDA_DEVICE_4SUBTRACT))
        PixA_REF_GPU, PixA_SCI_GPU = _t(PixA_REF_GPU, dev), _t(PixA_SCI_GPU, dev)
        PixA_mREF_GPU, PixA_mSCI_GPU = _t(PixA_mREF_GPU, dev), _t(PixA_mSCI_GPU, dev)

        # * assertions (PureCupyCustomizedPacket.py:105-118)
        assert torch.sum(torch.isnan(PixA_mREF_GPU)) == 0, "The masked reference image contains NaNs!"
        assert torch.sum(torch.isnan(PixA_mSCI_GPU)) == 0, "The masked science image contains NaNs!"
        assert PixA_REF_GPU.ndim == 2, "The input PixA_REF_GPU is not two-dimensional!"
        assert PixA_SCI_GPU.ndim == 2, "The input PixA_SCI_GPU is not two-dimensional!"
        assert PixA_mREF_GPU.ndim == 2, "The input PixA_mREF_GPU is not two-dimensional!"
        assert PixA_mSCI_GPU.ndim == 2, "The input PixA_mSCI_GPU is not two-dimensional!"
        assert PixA_REF_GPU.dtype == torch.float64, "The array does not have dtype cp.float64!"
        assert PixA_SCI_GPU.dtype == torch.float64, "The array does not have dtype cp.float64!"
        assert PixA_mREF_GPU.dtype == torch.float64, "The array does not have dtype cp.float64!"
        assert PixA_mSCI_GPU.dtype == torch.float64, "The array does not have dtype cp.float64!"
        assert ForceConv in ['REF', 'SCI']
        ConvdSide = ForceConv
        KerHW = GKerHW
        torch.cuda.set_device(dev)

        NaNmask_GPU = union_nan_mask(torch, PixA_REF_GPU, PixA_SCI_GPU)

        if VERBOSE_LEVEL in [0, 1, 2]:
            print('MeLOn CheckPoint: TRIGGER Function Compilations of SFFT-SUBTRACTION!')
        Tcomp_start = time.time()
        NX, NY = PixA_REF_GPU.shape
        SFFTConfig = SingleSFFTConfigure.SSC(NX=NX, NY=NY, KerHW=KerHW, KerPolyOrder=KerPolyOrder,
                                             BGPolyOrder=BGPolyOrder, ConstPhotRatio=ConstPhotRatio,
                                             BACKEND_4SUBTRACT="Cupy", VERBOSE_LEVEL=VERBOSE_LEVEL,
                                             CUDA_DEVICE_4SUBTRACT=dev.index)
        if VERBOSE_LEVEL in [1, 2]:
            print('\nMeLOn Report: Function Compilations of SFFT-SUBTRACTION TAKES [%.3f s]' % (time.time() - Tcomp_start))

        PixA_I_GPU, PixA_J_GPU, PixA_mI_GPU, PixA_mJ_GPU = assign_roles(
            torch, PixA_REF_GPU, PixA_SCI_GPU, PixA_mREF_GPU, PixA_mSCI_GPU, ConvdSide, NaNmask_GPU)

        Tsub_start = time.time()
        Solution_GPU, PixA_DIFF_GPU, _ = GeneralSFFTSubtract_PureCupy.GSS(
            PixA_I_GPU=PixA_I_GPU, PixA_J_GPU=PixA_J_GPU, PixA_mI_GPU=PixA_mI_GPU, PixA_mJ_GPU=PixA_mJ_GPU,
            SFFTConfig=SFFTConfig, ContamMask_I_GPU=None, VERBOSE_LEVEL=VERBOSE_LEVEL)
        if VERBOSE_LEVEL in [1, 2]:
            print('\nMeLOn Report: SFFT-SUBTRACTION TAKES [%.3f s]' % (time.time() - Tsub_start))

        PixA_DIFF_GPU = finish_diff(PixA_DIFF_GPU, ConvdSide, NaNmask_GPU)
        return Solution_GPU, PixA_DIFF_GPU
