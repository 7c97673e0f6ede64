"""Do the forbidden (removed) entries of the solution come back as exact zeros when the plan's buffers sit on recycled memory?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfft_amd.plan import Plan
from sfft_amd.utils.synthetic import make_pair
dev = torch.device("cuda", 0)
N0, N1, w = 4096, 4096, 8
pair = make_pair(512, 512, seed=3, mask=True)
g = {k: torch.from_numpy(np.tile(v, (8, 8))).to(dev).contiguous() for k, v in pair.items()}
for trial in range(3):
    junk = [torch.full((1 << 24,), float("nan"), dtype=torch.float64, device=dev) for _ in range(40)]   # 5 GB of NaN
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()
    plan = Plan(N0, N1, w, 2, 2, True, device=0)
    NEQ = plan.NEQ
    cen = 8 * 17 + 8
    forb = [ij * 289 + cen for ij in range(1, 6)]
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for rep in range(4):
            sol = torch.full((NEQ,), float("nan"), dtype=torch.float64, device=dev)
            diff = torch.empty((N0, N1), dtype=torch.float64, device=dev)
            plan.subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"], out_solution=sol, out_diff=diff)
            torch.cuda.synchronize()
            v = sol.cpu().numpy()
            print("trial", trial, "rep", rep, "graph", plan.query("SOLVE_GRAPH"), "forbidden:", v[forb], "nan elsewhere:", int(np.isnan(np.delete(v, forb)).sum()))
    plan.close()
