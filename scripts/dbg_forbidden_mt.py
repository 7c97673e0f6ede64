"""Forbidden entries of the Solution under concurrency: 4 host threads, one plan + stream each, outputs prefilled with NaN."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from sfft_amd.plan import Plan
from sfft_amd.utils.synthetic import make_pair
dev = torch.device("cuda", 0)
N0, N1, w = 4096, 4096, 8
pair = make_pair(512, 512, seed=3, mask=True)
g = {k: torch.from_numpy(np.tile(v, (8, 8))).to(dev).contiguous() for k, v in pair.items()}
junk = [torch.full((1 << 24,), float("nan"), dtype=torch.float64, device=dev) for _ in range(60)]
torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
S = 4
plans = [Plan(N0, N1, w, 2, 2, True, device=0) for _ in range(S)]
streams = [torch.cuda.Stream(dev) for _ in range(S)]
NEQ = plans[0].NEQ
forb = [ij * 289 + 8 * 17 + 8 for ij in range(1, 6)]
bad = [0] * S
ref = [None]
def worker(wi):
    torch.cuda.set_device(0)
    with torch.cuda.stream(streams[wi]):
        for rep in range(60):
            sol = torch.full((NEQ,), float("nan"), dtype=torch.float64, device=dev)
            diff = torch.empty((N0, N1), dtype=torch.float64, device=dev)
            plans[wi].subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"], out_solution=sol, out_diff=diff)
            v = sol.cpu().numpy()
            if not np.all(v[forb] == 0.0):
                bad[wi] += 1
                if bad[wi] <= 2: print("worker", wi, "rep", rep, "forbidden", v[forb], flush=True)
th = [threading.Thread(target=worker, args=(i,)) for i in range(S)]
[t.start() for t in th]; [t.join() for t in th]
print("SOL_MEMSET", os.environ.get("SFFT_SOL_MEMSET"), "NO_GRAPH", os.environ.get("SFFT_NO_GRAPH"), "bad per worker:", bad)
