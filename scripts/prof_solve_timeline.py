#!/usr/bin/env python3
"""Timeline of the LAST dense solve in a rocprofv3 kernel trace: every chol_* dispatch with its start (us after chol_begin), duration and
queue -- shows what the look-ahead of the outer-blocked factorisation overlaps.  usage: prof_solve_timeline.py <dir with *_results.db>"""
import glob, sqlite3, sys
dbs = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)
c = sqlite3.connect(dbs[0])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
rows = list(c.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
last = max(i for i, r in enumerate(rows) if "chol_begin" in r[0])
t0 = rows[last][1]
for r in rows[last:]:
    nm = r[0].split("(")[0].replace("void ", "")
    if not (nm.startswith("chol_") or nm.startswith("scatter")): 
        if nm.startswith("rows_") or nm.startswith("vconv"): print("   (%-24s start %8.1f dur %7.1f)" % (nm[:24], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3))
        continue
    print("%-22s start %8.1f  dur %7.1f  end %8.1f  q %s" % (nm[:22], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3, r[3] if qcol else "-"))
    if nm.startswith("scatter"): break
