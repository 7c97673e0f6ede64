#!/usr/bin/env python3
"""Print the sequence of kernel dispatches (name, duration us, gap to previous us) for one window of a rocprofv3 db."""
import glob, sqlite3, sys
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0; hi = int(sys.argv[4]) if len(sys.argv) > 4 else 80
dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
c = sqlite3.connect(dbs[0])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "select name, start, end from kernels order by start"
rows = list(c.execute(q))
prev_end = None; n = 0
for name, st, en in rows:
    nm = name.split("(")[0].replace("void ", "")
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    prev_end = en
    if pat and pat not in nm: continue
    if lo <= n < hi:
        print("%-28s dur %8.1f us   gap_before %7.1f us" % (nm[:28], (en - st) / 1e3, gap))
    n += 1
