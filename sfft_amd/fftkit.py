"""Real 2-D FFT handle over the C ABI (sfft_fft2_r2c / sfft_ifft2_c2r) and the spectrum-arithmetic entry points.
Used by the post-subtraction utilities (sfft_amd/utils/PureCupyFFTKits.py, *DeCorrelationCalculator.py)."""
import collections
import ctypes

import torch

from . import _lib

_S = lambda dev: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class FFTPlan:
    def __init__(self, N0, N1, device=0):
        self._h = ctypes.c_void_p()
        self.N0, self.N1, self.Nh, self.device = int(N0), int(N1), int(N1) // 2 + 1, int(device)
        _lib.check(_lib.lib().sfft_fft_plan_create(ctypes.byref(self._h), self.N0, self.N1, self.device))
        self.dev = torch.device("cuda", self.device)

    def rfft2(self, x, scale=1.0):
        """[N0][N1/2+1] complex128 = scale * DFT2(x), x real float64 (numpy.fft.rfft2 layout)."""
        x = x.to(device=self.dev, dtype=torch.float64).contiguous()
        assert tuple(x.shape) == (self.N0, self.N1)
        out = torch.empty((self.N0, self.Nh), dtype=torch.complex128, device=self.dev)
        _lib.check(_lib.lib().sfft_fft2_r2c(self._h, x.data_ptr(), out.data_ptr(), float(scale), _S(self.dev)))
        return out

    def irfft2(self, spec, scale=None):
        """real [N0][N1] from a half spectrum; default scale 1/(N0*N1) = numpy.fft.irfft2."""
        spec = spec.to(device=self.dev, dtype=torch.complex128).contiguous()
        assert tuple(spec.shape) == (self.N0, self.Nh)
        out = torch.empty((self.N0, self.N1), dtype=torch.float64, device=self.dev)
        sc = 1.0 / (self.N0 * self.N1) if scale is None else float(scale)
        _lib.check(_lib.lib().sfft_ifft2_c2r(self._h, spec.data_ptr(), out.data_ptr(), sc, _S(self.dev)))
        return out

    def close(self):
        if self._h:
            _lib.lib().sfft_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_CACHE = collections.OrderedDict()


def get_fft_plan(N0, N1, device=0):
    key = (int(device), int(N0), int(N1))
    p = _CACHE.get(key)
    if p is None:
        p = FFTPlan(N0, N1, device)
        _CACHE[key] = p
        while len(_CACHE) > 4:
            _CACHE.popitem(last=False)
    else:
        _CACHE.move_to_end(key)
    return p


def abs2_accumulate(acc, a, coeff, b=None):
    """acc += coeff * |a|^2 * |b|^2 (b optional); acc float64, a / b complex128, same shape."""
    assert acc.dtype == torch.float64 and a.dtype == torch.complex128 and acc.is_contiguous() and a.is_contiguous()
    bp = 0 if b is None else b.contiguous().data_ptr()
    _lib.check(_lib.lib().sfft_spec_abs2_accumulate(a.data_ptr(), bp, float(coeff), acc.data_ptr(), acc.numel(), _S(acc.device)))


def rsqrt(acc):
    out = torch.empty_like(acc)
    _lib.check(_lib.lib().sfft_real_rsqrt(acc.data_ptr(), out.data_ptr(), acc.numel(), _S(acc.device)))
    return out


def spec_multiply(a, b):
    """a * b elementwise; a complex128, b complex128 or float64."""
    a = a.contiguous(); b = b.contiguous()
    out = torch.empty_like(a)
    _lib.check(_lib.lib().sfft_spec_multiply(a.data_ptr(), b.data_ptr(), 1 if b.dtype == torch.float64 else 0, out.data_ptr(),
                                             a.numel(), _S(a.device)))
    return out


def half_to_full_real(half, N1):
    N0 = half.shape[0]
    out = torch.empty((N0, N1), dtype=torch.float64, device=half.device)
    _lib.check(_lib.lib().sfft_half_to_full_real(half.contiguous().data_ptr(), out.data_ptr(), int(N0), int(N1), _S(half.device)))
    return out
