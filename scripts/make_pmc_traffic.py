#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json from two pmc_summary.py outputs (FETCH_SIZE, WRITE_SIZE; KB per launch).

    python scripts/make_pmc_traffic.py profiles/rXX_pmc_FETCH_SIZE.txt profiles/rXX_pmc_WRITE_SIZE.txt > profiles/pmc_traffic.json

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE x2 for 16-byte-per-lane streaming reads
on gfx950; WRITE_SIZE as is.  Keys are the stage names bench.py uses; kernels launched several times per pair with
different plane counts (the FFT passes) are scaled from the mean launch to the launch bench.py times by the ratio of
algorithmic bytes (4096^2, KerHW 8, orders 2/2: Fij = 6)."""
import json
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


fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
Fij = 6
spec, img = 4096 * 2049 * 16, 4096 * 4096 * 8


def entry(kernels, scale=1.0, note=None):
    # a listed name matches a profiled kernel exactly or as the part before its template arguments ("greek_g1_mfma4g" -> "greek_g1_mfma4g<false, true>")
    def hit(name, k):
        return name == k or name.startswith(k + "<")
    f = sum(v[1] for name, v in fetch.items() if any(hit(name, k) for k in kernels))
    w = sum(v[1] for name, v in write.items() if any(hit(name, k) for k in kernels))
    e = {"kernels": kernels, "fetch_kb_raw": f, "write_kb": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0 * scale}
    if note:
        e["note"] = note
    return e


doc = {
    "_about": "HBM-side bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units) over "
              "`bench.py --streams 1 --steps 3 --warmup 1` on MI355X, 4096^2 KerHW 8 orders 2/2. FETCH_SIZE x2 (16-byte-per-lane "
              "streaming reads on gfx950, MI355X_MICROARCH.md), WRITE_SIZE unchanged (calibration: rows_c2r_diff_4096 writes "
              "131072.0 KB = exactly the 134217728-byte DIFF image).",
    "source_files": [sys.argv[1].split("/")[-1], sys.argv[2].split("/")[-1]],
    # cols_fwd_weighted_4096_q: one launch per pair (solve pass: 4 stage planes in, 7 planes out); the apply pass has no column transform
    "fwd_cols": entry(["cols_fwd_weighted_4096_q"]),
    # rows_r2c_4096: solve launch (2 images in, 4 stage planes out) vs apply launch (1 in, 3 out)
    "fwd_rows": entry(["rows_r2c_4096"], (2 * img + 4 * spec) / (0.5 * (2 * img + 4 * spec + img + 3 * spec)),
                      "mean over the solve and apply launches scaled to the solve launch"),
    "greek_g1": entry(["greek_g1_mfma4g", "greek_g1_mfma<2, false>", "greek_g1_mfma4<2>"]),
    "greek_g1b": entry(["greek_g1<8, 2>", "greek_g1_row0", "row_moments<5>", "gamma_rows", "gamma_patches"]),
    "construct": entry(["vconv_mixed2<2, 8, 4>", "vconv_mixed<2, 8, 4>", "vconv_direct", "kernel_ctab_mixed"]),
}
print(json.dumps(doc, indent=1))
