"""Development aid: one 4096^2 config-2 solve with SFFT_DF_TRACE=1; prints the per-step critical path of chol_dataflow (microseconds)."""
import os, sys, re, subprocess
if len(sys.argv) < 2:
    env = dict(os.environ, SFFT_DF_TRACE="1", SFFT_NO_GRAPH="1")
    out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    rows = [list(map(int, m.group(1).split())) for m in re.finditer(r"df_trace j=\d+ ([-\d ]+)", out.stderr)]
    rows = rows[-28:]
    names = ["acc_done", "diag_seen", "acquired", "diag_loaded", "trsm", "X_published", "ready2factor", "factored", "D_published", "chol_done", "border_out", "k_start", "k_end"]
    print(out.stdout[-300:])
    print("j   " + " ".join("%12s" % n for n in names) + "   step_us")
    prev = None
    for j, r in enumerate(rows):
        us = [v / 100.0 for v in r]
        print("%-3d " % j + " ".join("%12.2f" % v for v in us) + ("   %.2f" % (us[8] - prev) if prev is not None else ""))
        prev = us[8]
else:
    sys.path.insert(0, ".")
    import numpy as np, torch
    from sfft_amd.plan import Plan
    N = 4096
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    I = torch.randn((N, N), dtype=torch.float64, device=dev, generator=g)
    J = 1.2 * I + 0.1 * torch.roll(I, 1, 0) + torch.randn((N, N), dtype=torch.float64, device=dev, generator=g) * 0.1
    plan = Plan(N, N, 8, 2, 2, True, device=0)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            sol = plan.solve(I, J)
    torch.cuda.synchronize()
    print("solver", plan.query("LAST_SOLVER"), float(sol[8 * 17 + 8]) / N / N)
