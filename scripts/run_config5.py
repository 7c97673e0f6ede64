"""BASELINE config 5: one 9232 x 9216 (Roman WFI SCA sized) pair, KerHW 12, polynomial orders 3/3, fp64.  Times the GSS
(solve on the masked pair + apply) with per-stage times; also usable for other shapes:  python scripts/run_config5.py N0 N1 KerHW DK DB"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from sfft_amd.plan import Plan
from sfft_amd.utils.synthetic import make_pair
a = [int(v) for v in sys.argv[1:]]
N0, N1, w, DK, DB = (a + [9232, 9216, 12, 3, 3][len(a):])[:5]
dev = torch.device('cuda', 0)
pair = make_pair(N0, N1, seed=5, mask=True)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
I, J, mI, mJ = t(pair["REF"]), t(pair["SCI"]), t(pair["mREF"]), t(pair["mSCI"])
t0 = time.time()
plan = Plan(N0, N1, w, DK, DB, True, device=0)
torch.cuda.synchronize()
print("plan %dx%d w=%d DK=%d DB=%d: NEQ=%d, %.2f s, workspace %.1f GB" % (N0, N1, w, DK, DB, plan.NEQ, time.time() - t0, plan.query("WORKSPACE_BYTES") / 1e9))
plan.set_timing(True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    sol, diff = plan.subtract(I, J, mI, mJ)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("GSS %d: %.1f ms  stages %s solver %d" % (it, dt * 1e3, {k: round(v, 2) for k, v in plan.stage_ms().items()}, plan.query("LAST_SOLVER")))
d = diff.cpu().numpy()
print("rms(DIFF)=%.4f finite=%s  scaling=%.4f" % (np.sqrt(np.mean(d * d)), np.isfinite(d).all(), float(sol[w * (2 * w + 1) + w]) / N0 / N1))
