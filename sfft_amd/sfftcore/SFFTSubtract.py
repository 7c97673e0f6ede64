"""ESS / GSS -- mirror of sfft/sfftcore/SFFTSubtract.py:823-923 (host arrays) and :926-1450 (device arrays).

All numerics run in libsfft_amd.so; these classes only move data, check shapes with the reference's
error texts, and sequence solve/apply like the reference's two-pass GSS.
"""
import contextlib
import time

import numpy as np
import torch

_SFFT_BANNER = "\n  Saccadic Fast Fourier Transform (SFFT) algorithm -- MI355X/HIP backend (sfft_amd)\n"


def _plan(SFFTConfig):
    try:
        return SFFTConfig[1]['plan']
    except Exception:
        raise Exception('MeLOn ERROR: SFFTConfig was not produced by sfft_amd SingleSFFTConfigure.SSC')


@contextlib.contextmanager
def _bound_plan(SFFTConfig):
    """The config's plan, held for the duration of one ESS / GSS.  Plans are shared between configs of the same geometry
    (plan cache), so what belongs to the *config* -- the kernel-regularisation term of BSplineSFFT.SSC, which the reference
    derives from each config's own SFFTParam_dict at ESS time (BSplineSFFT.py:3293-3330) -- is put on the plan here, under
    the plan's lock, before anything is solved through this config."""
    plan = _plan(SFFTConfig)
    with plan.lock:
        if 'regularization' in SFFTConfig[1]:
            reg = SFFTConfig[1]['regularization']
            if plan._reg_token is not reg:
                plan.set_regularization(*reg)
                plan._reg_token = reg
        yield plan


def _as_device(x, plan, name):
    """float64, C-contiguous, on the plan's device.  Accepts torch tensors, numpy arrays and anything exposing
    __cuda_array_interface__ (e.g. CuPy arrays)."""
    dev = torch.device('cuda', plan.device)
    if isinstance(x, torch.Tensor):
        t = x
    elif hasattr(x, '__cuda_array_interface__'):
        t = torch.as_tensor(x, device=dev)
    else:
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
    return t.to(device=dev, dtype=torch.float64).contiguous()


class ElementalSFFTSubtract_PureCupy:
    @staticmethod
    def ESSPC(PixA_I_GPU, PixA_J_GPU, SFFTConfig, SFFTSolution_GPU=None, Subtract=False, VERBOSE_LEVEL=2):
        """Device arrays in, device arrays out (SFFTSubtract.py:926-1369)."""
        ta = time.time()
        plan = _plan(SFFTConfig)
        N0, N1 = SFFTConfig[0]['N0'], SFFTConfig[0]['N1']
        if VERBOSE_LEVEL in [1, 2]:
            print('\n --||--||--||--||-- TRIGGER SFFT SUBTRACTION --||--||--||--||-- ')
            print('\n ---||--- KerPolyOrder %d | BGPolyOrder %d | KerHW [%d] ---||--- '
                  % (SFFTConfig[0]['DK'], SFFTConfig[0]['DB'], SFFTConfig[0]['w0']))
        if tuple(PixA_I_GPU.shape) != (N0, N1) or tuple(PixA_J_GPU.shape) != (N0, N1):
            _error_message = 'INCONSISTENT shape of input images I & J, [%d, %d] required!' % (N0, N1)
            raise Exception('MeLOn ERROR: %s' % _error_message)
        I = _as_device(PixA_I_GPU, plan, 'PixA_I')
        J = _as_device(PixA_J_GPU, plan, 'PixA_J')
        with _bound_plan(SFFTConfig):
            if SFFTSolution_GPU is None:
                Solution_GPU = plan.solve(I, J)
            else:
                Solution_GPU = _as_device(SFFTSolution_GPU, plan, 'SFFTSolution')
            PixA_DIFF_GPU = None
            if Subtract:
                PixA_DIFF_GPU = plan.apply(I, J, Solution_GPU)
        if VERBOSE_LEVEL in [1, 2]:
            torch.cuda.synchronize(plan.device)
            print('\nMeLOn CheckPoint: SFFT-SUBTRACTION takes [%.4fs]' % (time.time() - ta))
            print('\n --||--||--||--||-- EXIT SFFT SUBTRACTION --||--||--||--||-- ')
        return Solution_GPU, PixA_DIFF_GPU


class ElementalSFFTSubtract:
    @staticmethod
    def ESS(PixA_I, PixA_J, SFFTConfig, SFFTSolution=None, Subtract=False,
            BACKEND_4SUBTRACT='Cupy', NUM_CPU_THREADS_4SUBTRACT=8, VERBOSE_LEVEL=2):
        """Host arrays in, host arrays out (SFFTSubtract.py:823-837; H2D :94-102, D2H :412, :461)."""
        if BACKEND_4SUBTRACT not in ('Cupy', 'HIP'):
            raise Exception("MeLOn ERROR: sfft_amd only provides the GPU backend (BACKEND_4SUBTRACT='Cupy')")
        sol_in = None if SFFTSolution is None else np.asarray(SFFTSolution, dtype=np.float64)
        Solution_GPU, PixA_DIFF_GPU = ElementalSFFTSubtract_PureCupy.ESSPC(
            PixA_I, PixA_J, SFFTConfig, SFFTSolution_GPU=sol_in, Subtract=Subtract, VERBOSE_LEVEL=VERBOSE_LEVEL)
        Solution = Solution_GPU.cpu().numpy()
        PixA_DIFF = None if PixA_DIFF_GPU is None else PixA_DIFF_GPU.cpu().numpy()
        return Solution, PixA_DIFF


class GeneralSFFTSubtract_PureCupy:
    @staticmethod
    def GSS(PixA_I_GPU, PixA_J_GPU, PixA_mI_GPU, PixA_mJ_GPU, SFFTConfig, ContamMask_I_GPU=None, VERBOSE_LEVEL=2):
        """Solve on the masked pair, apply to the full pair (SFFTSubtract.py:1371-1450)."""
        if VERBOSE_LEVEL in [2]:
            print(_SFFT_BANNER)
        # * Size-Check (the reference lists mI twice and never checks mJ, SFFTSubtract.py:1420; all four are checked here)
        tmplst = [tuple(PixA_I_GPU.shape), tuple(PixA_J_GPU.shape), tuple(PixA_mI_GPU.shape), tuple(PixA_mJ_GPU.shape)]
        if len(set(tmplst)) > 1:
            raise Exception('MeLOn ERROR: Input images should have same size!')
        plan = _plan(SFFTConfig)
        N0, N1 = SFFTConfig[0]['N0'], SFFTConfig[0]['N1']
        if tmplst[0] != (N0, N1):
            raise Exception('MeLOn ERROR: INCONSISTENT shape of input images I & J, [%d, %d] required!' % (N0, N1))
        I = _as_device(PixA_I_GPU, plan, 'PixA_I')
        J = _as_device(PixA_J_GPU, plan, 'PixA_J')
        mI = _as_device(PixA_mI_GPU, plan, 'PixA_mI')
        mJ = _as_device(PixA_mJ_GPU, plan, 'PixA_mJ')
        with _bound_plan(SFFTConfig):
            Solution_GPU, PixA_DIFF_GPU = plan.subtract(I, J, mI, mJ)
            # * Identify propagated contamination region through convolving I (SFFTSubtract.py:907-921, 1432-1448):
            #   apply the kernel-only solution (b_pq = 0) to the mask with J = 0 and threshold the result.
            ContamMask_CI_GPU = None
            if ContamMask_I_GPU is not None:
                Fpq = SFFTConfig[0]['Fpq']
                tSolution = Solution_GPU.clone()
                tSolution[-Fpq:] = 0.0
                _tmpI = _as_device(ContamMask_I_GPU, plan, 'ContamMask_I')
                _tmpJ = torch.zeros_like(J)
                _tmpD = plan.apply(_tmpI, _tmpJ, tSolution)
                FTHRESH = -0.001  # Emperical (reference value)
                ContamMask_CI_GPU = _tmpD < FTHRESH
        return Solution_GPU, PixA_DIFF_GPU, ContamMask_CI_GPU


class GeneralSFFTSubtract:
    @staticmethod
    def GSS(PixA_I, PixA_J, PixA_mI, PixA_mJ, SFFTConfig, ContamMask_I=None,
            BACKEND_4SUBTRACT='Cupy', NUM_CPU_THREADS_4SUBTRACT=8, VERBOSE_LEVEL=2):
        """Host arrays in, host arrays out (SFFTSubtract.py:839-923)."""
        if BACKEND_4SUBTRACT not in ('Cupy', 'HIP'):
            raise Exception("MeLOn ERROR: sfft_amd only provides the GPU backend (BACKEND_4SUBTRACT='Cupy')")
        Solution_GPU, PixA_DIFF_GPU, ContamMask_CI_GPU = GeneralSFFTSubtract_PureCupy.GSS(
            PixA_I, PixA_J, PixA_mI, PixA_mJ, SFFTConfig, ContamMask_I_GPU=ContamMask_I, VERBOSE_LEVEL=VERBOSE_LEVEL)
        Solution = Solution_GPU.cpu().numpy()
        PixA_DIFF = PixA_DIFF_GPU.cpu().numpy()
        ContamMask_CI = None if ContamMask_CI_GPU is None else ContamMask_CI_GPU.cpu().numpy()
        return Solution, PixA_DIFF, ContamMask_CI
