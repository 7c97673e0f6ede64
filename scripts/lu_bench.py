#!/usr/bin/env python3
"""Time the dense solvers alone (sfft_dbg_solve_dense) on a seeded SPD system of a plan's size: LU with partial pivoting (lu.hpp)
beside the Cholesky path.  `python scripts/lu_bench.py [w DK DB cpr]...` -- default: config 2's system (n = 1735) and n = 7300.
Under rocprofv3 --kernel-trace --stats this gives the per-kernel table of the LU chain."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfft_amd.plan import Plan

geoms = [(8, 2, 2, 1), (13, 3, 3, 0)]
if len(sys.argv) > 1:
    v = [int(x) for x in sys.argv[1:]]
    geoms = [tuple(v[i:i + 4]) for i in range(0, len(v), 4)]
dev = torch.device("cuda", 0)
for (w, DK, DB, cpr) in geoms:
    side = max(64, 4 * (2 * w + 1))
    plan = Plan(side, side, w, DK, DB, bool(cpr), device=0)
    n = plan.query("SOLVER_N")
    g = torch.Generator(device=dev)
    g.manual_seed(n)
    G = torch.randn((n, n + 64), dtype=torch.float64, device=dev, generator=g)
    A = G @ G.T / n + 0.5 * torch.eye(n, dtype=torch.float64, device=dev)
    b = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    del G
    plan.set_timing(True)
    out = {}
    for name, use_lu in (("cholesky", False), ("lu", True)):
        ms, wall = [], []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x = plan.solve_dense(A, b, use_lu=use_lu)
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(plan.stage_ms()["solve"])
        r = float((A @ x - b).abs().max() / (A.abs().max() * x.abs().max() * n))
        out[name] = (sorted(ms[1:])[len(ms[1:]) // 2], min(wall[1:]), r)
    print("n = %5d  cholesky %8.3f ms  lu %8.3f ms  ratio %5.2f   (residuals %.1e / %.1e)" % (n, out["cholesky"][0], out["lu"][0], out["lu"][0] / out["cholesky"][0], out["cholesky"][2], out["lu"][2]), flush=True)
    plan.close()
    if os.environ.get("LU_BENCH_VENDOR", "1") != "0":
        # yardstick, not product: what the platform's own dense solvers (torch.linalg on ROCm: rocSOLVER / hipSOLVER / MAGMA behind it) take for
        # the SAME system on this GPU -- the reference's solver is the CUDA counterpart of the first line (cupy.linalg.solve = getrf + getrs)
        def timed(fn, reps=6):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                y = fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return sorted(ts[1:])[len(ts[1:]) // 2], y
        t_lu, x1 = timed(lambda: torch.linalg.solve(A, b))
        t_ch, x2 = timed(lambda: torch.cholesky_solve(b[:, None], torch.linalg.cholesky_ex(A)[0])[:, 0])
        r1 = float((A @ x1 - b).abs().max() / (A.abs().max() * x1.abs().max() * n))
        r2 = float((A @ x2 - b).abs().max() / (A.abs().max() * x2.abs().max() * n))
        print("           torch.linalg.solve (vendor LU) %8.3f ms   torch cholesky + cholesky_solve %8.3f ms   (residuals %.1e / %.1e)" % (t_lu, t_ch, r1, r2), flush=True)
