#!/usr/bin/env python3
"""Absolute timeline (start / end in us relative to the first listed kernel) of the kernels of ONE image pair from a rocprofv3 --kernel-trace
database: which kernels overlap which (side-stream work beside the main stream).  usage: prof_timeline.py <dir or db> <index of the pair, default 3>
A pair is delimited by its `fill_system` launch: the window runs from the previous pair's scatter_solution to this pair's."""
import glob, sqlite3, sys
path = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
c = sqlite3.connect(dbs[0])
rows = [(n.split("(")[0].replace("void ", ""), s, e) for n, s, e in c.execute("select name, start, end from kernels order by start")]
marks = [i for i, r in enumerate(rows) if r[0].startswith("rows_r2c") or r[0].startswith("pack_rows")]
# the first row pass after a scatter_solution starts a pair
starts = [i for k, i in enumerate(marks) if k == 0 or any(r[0].startswith("scatter_solution") for r in rows[marks[k - 1]:i])]
lo = starts[min(which, len(starts) - 2)]; hi = starts[min(which, len(starts) - 2) + 1]
t0 = rows[lo][1]
for n, s, e in rows[lo:hi]:
    print("%-34s %9.1f -> %9.1f us  (%7.1f)" % (n[:34], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
