#!/usr/bin/env python3
"""List individual dispatches of kernels whose name contains a pattern from a rocprofv3 results database:
   python scripts/prof_kernel_list.py <dir-or-db> <pattern> [count]"""
import glob, sqlite3, sys
path, pat = sys.argv[1], sys.argv[2]
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
c = sqlite3.connect(dbs[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
q = "select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id where s.kernel_name like ? order by d.start" % (kd, ks)
rows = list(c.execute(q, ("%" + pat + "%",)))
print(len(rows), "dispatches")
for n, s, e, gx, gy, wx in rows[-cnt:]:
    print("%-40s %10.1f us  grid %d x %d  wg %d" % (n.split("(")[0][:40], (e - s) / 1e3, gx // max(wx, 1), gy, wx))
