#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json from a scripts/gpu_round.sh output directory:

    python scripts/make_pmc_traffic.py gpurun_out/r04 > profiles/pmc_traffic.json

For each config directory cfg2 / cfg3 / cfg5 it reads pmc_FETCH_SIZE.txt and pmc_WRITE_SIZE.txt (scripts/pmc_summary.py: mean KB per
launch and launch count per kernel) and bench_streams1_full.json (whose `stage_kernels` says which kernels each stage launched, as
recorded by the library) and writes, per config and stage, the HBM-side bytes of that stage for ONE image pair:
    sum over the stage's kernels of (2 x FETCH_SIZE + WRITE_SIZE) x launches / pairs profiled.
Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE x2 for 16-byte-per-lane streaming reads on gfx950;
WRITE_SIZE as is (calibration: rows_c2r_diff_4096 writes 131072.0 KB = exactly the 134217728-byte DIFF image).  A kernel that serves
two stages (the row pass: fwd_rows of the solve and prelim_apply of the apply) is split between them by their algorithmic bytes."""
import json
import os
import re
import sys


def parse(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        parts = line.split()
        if len(parts) < 4:
            continue
        name = " ".join(parts[:-3])
        out[name] = (int(parts[-2]), float(parts[-1]))
    return out


def base(name):
    return re.sub(r"\s+", "", name.replace("void ", "").split("<")[0].split("(")[0])


root = sys.argv[1]
doc = {"_about": "HBM-side bytes per stage and image pair from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB) over "
                 "`bench.py --config C --streams 1 --batch 4 --steps 2 --warmup 1`; FETCH_SIZE x2 (16-byte-per-lane streaming reads on gfx950, "
                 "MI355X_MICROARCH.md), WRITE_SIZE as is; stage -> kernels from the library's own record (sfft_stage_kernels); made by "
                 "scripts/make_pmc_traffic.py from " + os.path.basename(root.rstrip("/"))}
for cfg in ("2", "3", "5"):
    d = os.path.join(root, "cfg" + cfg)
    try:
        fetch, write = parse(os.path.join(d, "pmc_FETCH_SIZE.txt")), parse(os.path.join(d, "pmc_WRITE_SIZE.txt"))
        full = json.load(open(os.path.join(d, "bench_streams1_full.json")))
    except Exception as e:
        doc[cfg] = {"error": str(e)}
        continue
    sk = full["stage_kernels"]
    ab = {k: v["alg_bytes"] for k, v in full.get("hbm_stages", {}).items()}
    # bytes per kernel base name summed over all launches of the profiled run
    tot, launches = {}, {}
    for tab, mul in ((fetch, 2.0), (write, 1.0)):
        for name, (n, mean_kb) in tab.items():
            b = base(name)
            tot[b] = tot.get(b, 0.0) + mul * mean_kb * 1024.0 * n
            launches[b] = max(launches.get(b, 0), n)
    # pairs the profiled run subtracted: fill_system runs once per pair
    pairs = max(1, launches.get("fill_system", 1))
    users = {}
    for st, names in sk.items():
        if st == "prelim_solve":
            continue
        for n in names:
            users.setdefault(base(n), []).append(st)
    out = {"pairs_profiled": pairs}
    for st, names in sk.items():
        if st == "prelim_solve" or not names:
            continue
        total, used = 0.0, []
        for b in sorted(set(base(n) for n in names)):
            if b not in tot:
                continue
            share = 1.0
            sts = sorted(set(users[b]))
            if len(sts) > 1:          # split by algorithmic bytes when they are known for every user, else evenly
                share = ab[st] / sum(ab[s] for s in sts) if all(s in ab for s in sts) else 1.0 / len(sts)
            total += share * tot[b] / pairs
            used.append(b if share == 1.0 else "%s (%.2f of it)" % (b, share))
        if used:
            out[st] = {"kernels": used, "hbm_bytes_per_launch": total}
    doc[cfg] = out
print(json.dumps(doc, indent=1))
