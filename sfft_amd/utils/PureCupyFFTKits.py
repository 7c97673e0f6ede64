"""PureCupy_FFTKits -- mirror of sfft/utils/PureCupyFFTKits.py:35-105 on the HIP backend ("GPU arrays" = torch HIP tensors).
Padding / rolling / cropping are tensor plumbing; the transforms and the spectrum product run in libsfft_amd.so."""
import torch

from ..fftkit import get_fft_plan, spec_multiply

__all__ = ["PureCupy_FFTKits"]


class PureCupy_FFTKits:
    @staticmethod
    def KERNEL_CSZ(KERNEL_GPU, NX_IMG, NY_IMG, NORMALIZE_KERNEL=False):
        """Circular Shift the kernel and extend to the target size (:37-53)."""
        N0, N1 = NX_IMG, NY_IMG
        L0, L1 = KERNEL_GPU.shape
        W0, W1 = (L0 - 1) // 2, (L1 - 1) // 2
        assert L0 % 2 == 1 and L1 % 2 == 1     # only odd-sized kernels, like the reference
        K = KERNEL_GPU / torch.sum(KERNEL_GPU) if NORMALIZE_KERNEL else KERNEL_GPU
        TZP = torch.nn.functional.pad(K.to(torch.float64), (0, N1 - L1, 0, N0 - L0), mode='constant', value=0.)
        return torch.roll(TZP, shifts=(-W0, -W1), dims=(0, 1))

    @staticmethod
    def KERNEL_CSZ_INV(KIMG_GPU, NX_KERN, NY_KERN, VERBOSE_LEVEL=2):
        """Inverse Circular Shift the kernel and truncate to the target size (:55-71)."""
        L0, L1 = NX_KERN, NY_KERN
        W0, W1 = (L0 - 1) // 2, (L1 - 1) // 2
        assert L0 % 2 == 1 and L1 % 2 == 1
        KIMG_iCSZ_GPU = torch.roll(KIMG_GPU, shifts=(W0, W1), dims=(0, 1))
        KERNEL_GPU = KIMG_iCSZ_GPU[:L0, :L1]
        if VERBOSE_LEVEL in [1, 2]:
            LOSE_RATIO = 1. - torch.sum(torch.abs(KERNEL_GPU)) / torch.sum(torch.abs(KIMG_iCSZ_GPU))
            print("MeLOn CheckPoint: Kernel Truncation Loses APE = [%.4f %s] " % (float(LOSE_RATIO) * 100, '%'))
        return KERNEL_GPU

    @staticmethod
    def FFT_CONVOLVE(PixA_Inp_GPU, KERNEL_GPU, PAD_FILL_VALUE=0., NAN_FILL_VALUE=0., NORMALIZE_KERNEL=False,
                     FORCE_OUTPUT_C_CONTIGUOUS=False, FFT_BACKEND="Cupy"):
        """FFT convolution with a W-pixel constant border (:73-105): ifft2(fft2(padded image) * fft2(CSZ kernel)).real, cropped."""
        N0, N1 = PixA_Inp_GPU.shape
        L0, L1 = KERNEL_GPU.shape
        assert L0 % 2 == 1 and L1 % 2 == 1
        W0, W1 = (L0 - 1) // 2, (L1 - 1) // 2
        NX_IMG, NY_IMG = N0 + 2 * W0, N1 + 2 * W1
        E = torch.nn.functional.pad(PixA_Inp_GPU.to(torch.float64), (W1, W1, W0, W0), mode='constant', value=float(PAD_FILL_VALUE))
        if NAN_FILL_VALUE is not None:
            E = torch.nan_to_num(E, nan=float(NAN_FILL_VALUE), posinf=float('inf'), neginf=float('-inf'))
        KIMG = PureCupy_FFTKits.KERNEL_CSZ(KERNEL_GPU, NX_IMG, NY_IMG, NORMALIZE_KERNEL=NORMALIZE_KERNEL)
        plan = get_fft_plan(NX_IMG, NY_IMG, E.device.index)
        out = plan.irfft2(spec_multiply(plan.rfft2(E), plan.rfft2(KIMG)))
        out = out[W0: NX_IMG - W0, W1: NY_IMG - W1]
        if FORCE_OUTPUT_C_CONTIGUOUS and not out.is_contiguous():
            out = out.contiguous()
        return out
