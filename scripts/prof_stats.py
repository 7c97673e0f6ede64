#!/usr/bin/env python3
"""Print per-kernel stats (calls, total us, avg us, %) from a rocprofv3 results database."""
import glob
import sqlite3
import sys

path = sys.argv[1]
dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
c = sqlite3.connect(dbs[0])
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
print("%-64s %6s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for n, k, t, a, p in rows:
    if p < 0.02:
        continue
    name = n.split("(")[0].replace("void ", "")
    print("%-64s %6d %12.1f %10.1f %6.2f" % (name[:64], k, t, a, p))
