"""PureCupy_DeCorrelation_Calculator.PCDC -- mirror of sfft/utils/PureCupyDeCorrelationCalculator.py:46-126 on the HIP
backend.  Kernels are real, so their spectra are kept as half spectra; the denominator is real and conjugate-symmetric."""
import numpy as np
import torch

from ..fftkit import get_fft_plan, abs2_accumulate, rsqrt, half_to_full_real
from .PureCupyFFTKits import PureCupy_FFTKits

__all__ = ["PureCupy_DeCorrelation_Calculator"]


class PureCupy_DeCorrelation_Calculator:
    @staticmethod
    def PCDC(NX_IMG, NY_IMG, KERNEL_GPU_JQueue, BKGSIG_JQueue, KERNEL_GPU_IQueue=[], BKGSIG_IQueue=[],
             MATCH_KERNEL_GPU=None, REAL_OUTPUT=False, REAL_OUTPUT_SIZE=None, NORMALIZE_OUTPUT=True, VERBOSE_LEVEL=2,
             CUDA_DEVICE=None, DENO_CLIP_RATIO=None):
        """Decorrelation kernel: Fourier-space (REAL_OUTPUT=False, full [NX_IMG][NY_IMG] float64) or real-space.
        DENO_CLIP_RATIO (not in the reference's PCDC; used by BSpline_DeCorrelation.BDC, sfft/BSplineSFFT.py:4845-4847): floor the
        denominator at max / ratio."""
        NUM_I, NUM_J = len(KERNEL_GPU_IQueue), len(KERNEL_GPU_JQueue)
        assert NUM_J > 0
        if NUM_I == 0:
            if NUM_J < 2:
                raise Exception('MeLOn ERROR: %s' % 'IMAGE-STACKING MODE Requires at least 2 J-IMAGE!')
            if np.sum([K is not None for K in KERNEL_GPU_JQueue]) == 0:
                raise Exception('MeLOn ERROR: %s' % 'IMAGE-STACKING MODE Requires at least 1 non-None J-KERNEL!')
        if NUM_I >= 1:
            _Q = list(KERNEL_GPU_JQueue) + list(KERNEL_GPU_IQueue) + [MATCH_KERNEL_GPU]
            if np.sum([K is not None for K in _Q]) == 0:
                raise Exception('MeLOn ERROR: %s' % 'IMAGE-SUBTRACTION MODE Requires at least 1 non-None J/I/MATCH-KERNEL!')
        dev_idx = CUDA_DEVICE
        if dev_idx is None:
            anyk = [K for K in list(KERNEL_GPU_JQueue) + list(KERNEL_GPU_IQueue) + [MATCH_KERNEL_GPU] if K is not None][0]
            dev_idx = anyk.device.index if isinstance(anyk, torch.Tensor) and anyk.is_cuda else torch.cuda.current_device()
        dev = torch.device("cuda", dev_idx)
        plan = get_fft_plan(NX_IMG, NY_IMG, dev_idx)
        DELTA = torch.tensor([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=torch.float64, device=dev)

        def spectrum(K):
            K = DELTA if K is None else torch.as_tensor(K, device=dev).to(torch.float64)
            return plan.rfft2(PureCupy_FFTKits.KERNEL_CSZ(K, NX_IMG, NY_IMG))

        FDENO = torch.zeros((NX_IMG, NY_IMG // 2 + 1), dtype=torch.float64, device=dev)
        for K, BKGSIG in zip(KERNEL_GPU_JQueue, BKGSIG_JQueue):
            abs2_accumulate(FDENO, spectrum(K), float(BKGSIG) ** 2 / NUM_J ** 2)
        FMK = spectrum(MATCH_KERNEL_GPU)
        for K, BKGSIG in zip(KERNEL_GPU_IQueue, BKGSIG_IQueue):
            abs2_accumulate(FDENO, spectrum(K), float(BKGSIG) ** 2 / NUM_I ** 2, FMK)
        if DENO_CLIP_RATIO is not None:
            FDENO = torch.clamp_min(FDENO, float(FDENO.max()) / float(DENO_CLIP_RATIO))   # the half spectrum holds every distinct value
        FK_half = rsqrt(FDENO)                                  # 1 / sqrt(denominator): real & conjugate-symmetric
        if not REAL_OUTPUT:
            FKDECO = half_to_full_real(FK_half, NY_IMG)
            if NORMALIZE_OUTPUT:
                FKDECO = FKDECO * (1. / FKDECO[0, 0])
            return FKDECO
        assert REAL_OUTPUT_SIZE is not None
        KDECO = plan.irfft2(torch.complex(FK_half, torch.zeros_like(FK_half)))
        KDECO = PureCupy_FFTKits.KERNEL_CSZ_INV(KDECO, REAL_OUTPUT_SIZE[0], REAL_OUTPUT_SIZE[1], VERBOSE_LEVEL=VERBOSE_LEVEL)
        if NORMALIZE_OUTPUT:
            KDECO = KDECO * (1. / torch.sum(KDECO))
        return KDECO
