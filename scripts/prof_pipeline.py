#!/usr/bin/env python3
"""What the pipelined run looks like from the device: from a rocprofv3 --kernel-trace database of `bench.py` with several pairs in flight,
over a steady-state window (between the fill_system launches at 30 % and 80 % of all of them):
  - per kernel: launches, mean duration WHILE OVERLAPPED with other pairs' kernels (compare with the one-pair table of the same build),
    and its share of the window if it ran alone back to back (sum of durations / window);
  - the distribution of how many chip-filling kernels are in flight at once, and the fraction of the window with none.
usage: prof_pipeline.py <dir or .db> [pairs/s of that run]"""
import glob, sqlite3, sys, collections
path = sys.argv[1]
dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
c = sqlite3.connect(dbs[0])
rows = [(n.split("(")[0].replace("void ", ""), s, e) for n, s, e in c.execute("select name, start, end from kernels order by start")]
# steady state: from the launch of the fill_system at 30 % of all of them to the one at 80 % (set-up, warm-up and the one-pair legs behind the
# timed region stay outside)
fills = [s for n, s, e in rows if n.startswith("fill_system")]
a, b = fills[int(0.3 * len(fills))], fills[int(0.8 * len(fills))]
win = [(n, max(s, a), min(e, b)) for n, s, e in rows if e > a and s < b]
W = (b - a) / 1e3
BIG = ("rows_r2c", "cols_fwd", "greek_g1_mfma", "vconv", "rows_c2r", "greek_g2", "fill_system", "gamma_")
SOLVE = ("chol_dataflow",)
agg = collections.OrderedDict()
for n, s, e in win:
    d = agg.setdefault(n, [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
npairs = agg.get("fill_system", [1])[0]
print("window %.1f us, %d pairs -> %.1f us per pair (%.1f pairs/s)" % (W, npairs, W / npairs, 1e6 * npairs / W))
print("%-36s %7s %10s %12s %9s" % ("kernel", "calls", "mean us", "us per pair", "x window"))
for n, (k, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-36s %7d %10.1f %12.1f %9.2f" % (n[:36], k, tot / k, tot / npairs, tot / W))
# concurrency of the chip-filling kernels (sweep)
ev = []
for n, s, e in win:
    kind = 1 if n.startswith(BIG) else (2 if n.startswith(SOLVE) else 0)
    if kind:
        ev.append((s, +1, kind)); ev.append((e, -1, kind))
ev.sort()
cur = {1: 0, 2: 0}; last = a; hist = collections.Counter(); hist_s = collections.Counter()
for t, dlt, kind in ev:
    hist[cur[1]] += t - last; hist_s[cur[2]] += t - last; last = t
    cur[kind] += dlt
hist[cur[1]] += b - last; hist_s[cur[2]] += b - last
print("chip-filling kernels in flight: " + ", ".join("%d: %.0f %%" % (k, 100.0 * v / (b - a)) for k, v in sorted(hist.items())))
print("dense solves in flight:         " + ", ".join("%d: %.0f %%" % (k, 100.0 * v / (b - a)) for k, v in sorted(hist_s.items())))
