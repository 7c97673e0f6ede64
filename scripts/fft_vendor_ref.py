"""Reference point for the forward-FFT kernels: torch.fft.rfft2 (rocFFT) on the same batch shape, MI355X.
Not used by the product; run on the GPU box:  python scripts/fft_vendor_ref.py"""
import torch, time
dev = torch.device("cuda", 0)
for batch in (1, 7):
    x = torch.randn(batch, 4096, 4096, dtype=torch.float64, device=dev)
    for _ in range(3):
        y = torch.fft.rfft2(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        y = torch.fft.rfft2(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    gb = batch * (4096 * 4096 * 8 + 2 * 4096 * 2049 * 16 + 4096 * 2049 * 16) / 1e9   # read image, write+read+write half spectrum
    print("rfft2 batch %d: %.3f ms  (%.1f us / plane, %.2f TB/s on the 2-pass minimum)" % (batch, ms, 1e3 * ms / batch, gb / ms))
    z = torch.randn(batch, 4096, 2049, dtype=torch.complex128, device=dev)
    for _ in range(3):
        w = torch.fft.fft(z, dim=1)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        w = torch.fft.fft(z, dim=1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("column c2c (dim=1, 4096 pts, 2049 columns) batch %d: %.3f ms (%.1f us / plane, %.2f TB/s)" % (batch, ms, 1e3 * ms / batch, batch * 2 * 4096 * 2049 * 16 / 1e9 / ms))
