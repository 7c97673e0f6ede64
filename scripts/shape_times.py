"""Time one GSS (solve + apply) for a list of image shapes: python scripts/shape_times.py 2048x4096:8:2 1024x1024:4:2 ..."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from sfft_amd.plan import get_plan
dev = torch.device('cuda', 0)
for spec in sys.argv[1:]:
    shp, w, dk = spec.split(':')
    N0, N1 = [int(v) for v in shp.split('x')]
    w, dk = int(w), int(dk)
    rng = np.random.default_rng(1)
    I = torch.from_numpy(rng.normal(size=(N0, N1)) + 10).to(dev)
    J = torch.from_numpy(rng.normal(size=(N0, N1)) + 12).to(dev)
    plan = get_plan(N0, N1, w, dk, dk, True, 0)
    plan.set_timing(True)
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.time()
        plan.subtract(I, J, I, J)
        torch.cuda.synchronize(); best = min(best, time.time() - t0)
    st = plan.stage_ms()
    print("%-14s w=%d DK=%d  %.2f ms  %s" % (shp, w, dk, best * 1e3, {k: round(v, 2) for k, v in st.items()}))
