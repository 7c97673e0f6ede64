#!/usr/bin/env python3
"""Yardstick, not product: the library's real 2-D transforms (sfft_fft2_r2c / sfft_ifft2_c2r: the same row / column passes the
subtraction pipeline runs) beside the platform's own FFT (torch.fft.rfft2 / irfft2 = rocFFT) on the same fp64 images and the same GPU.
Algorithmic bytes of one transform: image (8 P) + half spectrum (16 N0 Nh) read or written once per pass = 8 P + 3 x 16 N0 Nh.
`python scripts/fft_vs_vendor.py [N0 N1]...` -- default: the image sizes of BASELINE configs 2, 3 and 5."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfft_amd.fftkit import FFTPlan

shapes = [(4096, 4096), (6144, 6144), (9232, 9216)]
if len(sys.argv) > 2:
    v = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(v[i:i + 2]) for i in range(0, len(v), 2)]
dev = torch.device("cuda", 0)


def timed(fn, reps=12):
    ts = []
    y = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        y = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], y


for (N0, N1) in shapes:
    g = torch.Generator(device=dev)
    g.manual_seed(N0 + N1)
    x = torch.randn((N0, N1), dtype=torch.float64, device=dev, generator=g)
    plan = FFTPlan(N0, N1, 0)
    Nh = N1 // 2 + 1
    out = torch.empty((N0, Nh), dtype=torch.complex128, device=dev)
    back = torch.empty((N0, N1), dtype=torch.float64, device=dev)
    import ctypes
    from sfft_amd import _lib
    S = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L = _lib.lib()
    t_f, _ = timed(lambda: _lib.check(L.sfft_fft2_r2c(plan._h, x.data_ptr(), out.data_ptr(), 1.0, S())))
    t_i, _ = timed(lambda: _lib.check(L.sfft_ifft2_c2r(plan._h, out.data_ptr(), back.data_ptr(), 1.0 / (N0 * N1), S())))
    v_f, ref = timed(lambda: torch.fft.rfft2(x))
    v_i, refb = timed(lambda: torch.fft.irfft2(ref, s=(N0, N1)))
    e_f = float((out - ref).abs().max() / ref.abs().max())
    e_i = float((back - x).abs().max() / x.abs().max())
    gb = (8.0 * N0 * N1 + 3 * 16.0 * N0 * Nh) / 1e9
    print("%5d x %5d  r2c: here %7.3f ms (%4.2f TB/s of algorithmic bytes)  rocFFT %7.3f ms   |  c2r: here %7.3f ms  rocFFT %7.3f ms   (max rel diff %.1e, round trip %.1e)"
          % (N0, N1, t_f, gb / t_f, v_f, t_i, v_i, e_f, e_i), flush=True)
    plan.close()
    del x, out, back, ref, refb
