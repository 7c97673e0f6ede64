"""ConvKernel_Convertion -- mirror of sfft/utils/ConvKernelConvertion.py:16-32 (host numpy helpers)."""
import numpy as np


class ConvKernel_Convertion:
    def CSZ(ConvKernel, N0, N1):
        """Circular-Shift & tail-Zero-padding of a small kernel to the image size."""
        L0, L1 = ConvKernel.shape
        w0, w1 = (L0 - 1) // 2, (L1 - 1) // 2
        TailZP = np.pad(ConvKernel, ((0, N0 - L0), (0, N1 - L1)), 'constant', constant_values=(0, 0))
        return np.roll(np.roll(TailZP, -w0, axis=0), -w1, axis=1)

    def iCSZ(KIMG, L0, L1, VERBOSE=True):
        w0, w1 = (L0 - 1) // 2, (L1 - 1) // 2
        KIMG_iCSZ = np.roll(np.roll(KIMG, w1, axis=1), w0, axis=0)
        ConvKernel = KIMG_iCSZ[:L0, :L1]
        if VERBOSE:
            lost_weight = 1.0 - np.sum(np.abs(ConvKernel)) / np.sum(np.abs(KIMG_iCSZ))
            print('MeLOn CheckPoint: Tail-Truncation Lost-Weight [%.4f %s] (Absolute Percentage Error) ' % (lost_weight * 100, '%'))
        return ConvKernel
