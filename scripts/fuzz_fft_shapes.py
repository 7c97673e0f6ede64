"""Randomised check of the 2-D transforms over many image shapes (on-chip radix-16 / mixed-radix / Bluestein axes, four-step
axes, panel planes): forward spectrum against numpy.fft, and one full subtraction against the Fourier-domain apply.
usage (GPU box): python scripts/fuzz_fft_shapes.py [count] [seed]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from sfft_amd.plan import Plan

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
MAXSIDE = int(os.environ.get("FUZZ_MAX_SIDE", "3000"))      # FUZZ_MAX_SIDE=12000: every side up to 12 000 must plan
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda', 0)
special = [8, 9, 12, 15, 16, 17, 24, 27, 31, 32, 33, 48, 63, 64, 65, 81, 96, 97, 127, 128, 129, 243, 255, 256, 257, 384, 511, 512, 513,
           729, 768, 1000, 1023, 1024, 1025, 1152, 1536, 2047, 2048, 2049, 2187, 3072, 4095, 4097, 4608, 5000, 6144, 8192, 8193, 9216]
worst = 0.0
unsupported = 0
if os.environ.get("FUZZ_PLAN_ONLY"):       # FUZZ_PLAN_ONLY=500: that many random sides in [8, MAXSIDE], as rows and as columns: every one must plan
    sides = sorted(set(int(x) for x in rng.integers(8, MAXSIDE + 1, size=int(os.environ["FUZZ_PLAN_ONLY"]))))
    bad = []
    for N in sides:
        for shape in ((N, 16), (16, N)):
            try:
                Plan(shape[0], shape[1], 1, 0, 0, True, device=0).close()
            except Exception as e:
                bad.append((shape, str(e)[:50]))
    print("%d distinct sides in [8, %d] planned as rows and as columns; refused: %d %s" % (len(sides), MAXSIDE, len(bad), bad[:5]))
    sys.exit(1 if bad else 0)
for it in range(count):
    N0 = int(rng.choice(special)) if rng.random() < 0.6 else int(rng.integers(8, MAXSIDE))
    N1 = int(rng.choice(special)) if rng.random() < 0.6 else int(rng.integers(8, MAXSIDE))
    if MAXSIDE > 3000 and rng.random() < 0.5:      # one long side, one short: the long axes are what is being fuzzed
        if rng.random() < 0.5: N1 = int(rng.integers(8, 64))
        else: N0 = int(rng.integers(8, 64))
    if N0 * N1 > 40e6:
        N1 = max(8, int(40e6 // N0))
    w = int(rng.integers(1, 4)); DK = int(rng.integers(0, 3))
    try:
        plan = Plan(N0, N1, w, DK, 1, True, device=0)
    except Exception as e:
        unsupported += 1
        print("%5d x %5d  unsupported: %s" % (N0, N1, str(e)[:60])); continue
    img = rng.normal(size=(N0, N1)) * 30 + 5
    ij = (int(rng.integers(0, DK + 1)), 0)
    F = plan.forward_spectrum(torch.from_numpy(img).to(dev), ij[0], ij[1]).cpu().numpy()
    cx = ((np.arange(N0) + 1.0) / N0)[:, None]
    ref = (np.fft.fft2(img * cx ** ij[0]) / (N0 * N1))[:, :N1 // 2 + 1]
    e1 = np.max(np.abs(F - ref)) / np.max(np.abs(ref))
    # one subtraction: default apply against the Fourier-domain apply (independent column transforms)
    J = rng.normal(size=(N0, N1)) * 30 + 9
    I_d, J_d = torch.from_numpy(img).to(dev), torch.from_numpy(J).to(dev)
    s1, d1 = plan.subtract(I_d, J_d, I_d, J_d)
    plan.close()
    os.environ["SFFT_NO_VCONV"] = "1"
    try:
        plan2 = Plan(N0, N1, w, DK, 1, True, device=0)
    finally:
        os.environ.pop("SFFT_NO_VCONV", None)
    s2, d2 = plan2.subtract(I_d, J_d, I_d, J_d)
    plan2.close()
    d1, d2 = d1.cpu().numpy(), d2.cpu().numpy()
    e2 = np.sqrt(np.mean((d1 - d2) ** 2)) / np.sqrt(np.mean(J ** 2))
    worst = max(worst, e1, e2)
    flag = "" if (e1 < 1e-12 and e2 < 1e-10) else "   <-- CHECK"
    print("%5d x %5d  w=%d DK=%d  spectrum %.1e  apply-vs-apply %.1e%s" % (N0, N1, w, DK, e1, e2, flag), flush=True)
print("worst", worst, " unsupported shapes:", unsupported, "of", count)
