"""DeCorrelation_Calculator.DCC -- mirror of sfft/utils/DeCorrelationCalculator.py:11-104 (host arrays in, host array out).
The transforms run on the GPU through PureCupy_DeCorrelation_Calculator; there is no CPU path in sfft_amd."""
import math

import numpy as np
import torch

from .PureCupyDeCorrelationCalculator import PureCupy_DeCorrelation_Calculator

__all__ = ["DeCorrelation_Calculator"]


class DeCorrelation_Calculator:
    @staticmethod
    def DCC(MK_JLst, SkySig_JLst, MK_ILst=[], SkySig_ILst=[], MK_Fin=None, KERatio=2.0, VERBOSE_LEVEL=2, CUDA_DEVICE=None):
        NumI, NumJ = len(MK_ILst), len(MK_JLst)
        if NumI == 0:
            Mode = 'Image-Stacking'
            if NumJ < 2:
                raise Exception('MeLOn ERROR: %s' % 'Image-Stacking Mode requires at least 2 J-images!')
            if np.sum([MKj is not None for MKj in MK_JLst]) == 0:
                raise Exception('MeLOn ERROR: %s' % 'Image-Stacking Mode requires at least 1 not-None J-kernel!')
        if NumI >= 1:
            Mode = 'Image-Subtraction'
            if NumJ == 0:
                raise Exception('MeLOn ERROR: %s' % 'Image-Subtraction Mode requires at least 1 I-image & 1 J-image!')
            if np.sum([MK is not None for MK in list(MK_JLst) + list(MK_ILst) + [MK_Fin]]) == 0:
                raise Exception('MeLOn ERROR: %s' % 'Image-Subtraction Mode requires at least 1 not-None J/I/Fin-kernel!')
        MK_Queue = list(MK_JLst)
        if Mode == 'Image-Subtraction':
            MK_Queue += [MK_Fin] + list(MK_ILst)
        L0_KDeCo = int(round(KERatio * np.max([MK.shape[0] for MK in MK_Queue if MK is not None])))
        L1_KDeCo = int(round(KERatio * np.max([MK.shape[1] for MK in MK_Queue if MK is not None])))
        if L0_KDeCo % 2 == 0: L0_KDeCo += 1
        if L1_KDeCo % 2 == 0: L1_KDeCo += 1
        if VERBOSE_LEVEL in [1, 2]:
            print('MeLOn CheckPoint: %s' % ('DeCorrelation Kernel with size [%d, %d]' % (L0_KDeCo, L1_KDeCo)))
        # trivial image size, just typically larger than the kernel size (DeCorrelationCalculator.py:63-65)
        N0 = 2 ** (math.ceil(np.log2(np.max([MK.shape[0] for MK in MK_Queue if MK is not None]))) + 1)
        N1 = 2 ** (math.ceil(np.log2(np.max([MK.shape[1] for MK in MK_Queue if MK is not None]))) + 1)
        dev = torch.device("cuda", torch.cuda.current_device() if CUDA_DEVICE is None else int(CUDA_DEVICE))
        tg = lambda K: None if K is None else torch.from_numpy(np.ascontiguousarray(K, dtype=np.float64)).to(dev)
        # stacking mode ignores MK_Fin like the reference (the I loop is empty)
        KDeCo = PureCupy_DeCorrelation_Calculator.PCDC(
            NX_IMG=N0, NY_IMG=N1, KERNEL_GPU_JQueue=[tg(K) for K in MK_JLst], BKGSIG_JQueue=list(SkySig_JLst),
            KERNEL_GPU_IQueue=[tg(K) for K in MK_ILst], BKGSIG_IQueue=list(SkySig_ILst), MATCH_KERNEL_GPU=tg(MK_Fin),
            REAL_OUTPUT=True, REAL_OUTPUT_SIZE=(L0_KDeCo, L1_KDeCo), NORMALIZE_OUTPUT=True,
            VERBOSE_LEVEL=VERBOSE_LEVEL, CUDA_DEVICE=dev.index)
        return KDeCo.cpu().numpy()
