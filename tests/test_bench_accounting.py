"""bench.py's algorithmic byte / flop figures (the numerators of `roofline`) against hand counts for BASELINE config 2, and the latest committed
bench line against the contract's keys.  No GPU."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_and_flops_of_config2():
    b = _bench()
    N0 = N1 = 4096
    w, Fij, Fpq = 8, 6, 6
    Nh = N1 // 2 + 1
    spec = 16 * N0 * Nh
    img = 8 * N0 * N1
    ab = b.alg_bytes(N0, N1, w, Fij, Fpq, 3, 2, True, theta_fused=True, omg_counts=(15, 6), decimated=True, chunks=4)
    assert ab["fwd_rows"] == 2 * img + 4 * spec                       # two masked images in, 3 + 1 stage planes out
    assert ab["fwd_cols"] == 4 * spec + 7 * spec                      # 4 stage planes in, Fij + 1 spectra out
    partial = 4 * 16 * (N1 // 2 + 4) * (21 * 33 + 6 * 17)             # 4 chunks x (21 products x 33 lags + 6 Theta passes x 17 lags) complex sums per column
    assert ab["greek_g1"] == 7 * spec + partial
    assert ab["construct"] == 3 * spec + spec and ab["prelim_apply"] == img + 3 * spec and ab["inverse"] == spec + 2 * img
    # direct pruned transform: per spectrum element a complex product (6, real product 3) and 4 real FMAs per lag pair
    direct = N0 * Nh * (15 * (6 + 8 * 16) + 6 * (3 + 4 * 16) + 6 * (6 + 8 * 8))
    assert ab["greek_g1_flops_direct"] == direct
    # decimated: products on every row, butterflies (4 / 2 additions per row pair) and the lag sums on half the rows
    dec = N0 * Nh * (15 * 6 + 6 * 3 + 6 * 6) + (N0 // 2) * Nh * (15 * (4 + 8 * 16) + 6 * (2 + 4 * 16) + 6 * (4 + 8 * 8))
    assert ab["greek_g1_flops"] == dec and dec < 0.56 * direct
    # 183 transforms of 4 passes over a complex plane + the element-wise Greek / preliminary passes + 5 real images
    assert ab["B_alg_reference"] == 183 * 4 * 16 * N0 * N1 + 156 * 16 * N0 * N1 + 13 * 16 * N0 * N1 + 5 * 8 * N0 * N1


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def _check_contract(d):
    for k in CONTRACT:
        assert k in d, k
    assert d["unit"] == "image-pairs/s" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "4096x4096" in d["config"]["workload"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"] + 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and d["value"] > 100 * c["value"]


def _last_full_line():
    p = os.path.join(ROOT, "profiles", "bench_last_full.json")
    if os.path.exists(p):
        return json.loads(open(p).read().strip().splitlines()[-1])
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert lines, "no committed default bench line under profiles/"
    return json.loads(open(lines[-1]).read().strip().splitlines()[-1])


def test_committed_bench_line_keeps_the_contract():
    _check_contract(_last_full_line())


def test_compact_line_is_small_and_keeps_the_contract():
    """Round 3's 26 KB line outgrew the driver's ~8 KB stdout tail and arrived unparseable: the LAST stdout line is now a compact
    object built from the full one (which goes to profiles/bench_last_full.json)."""
    b = _bench()
    full = _last_full_line()
    c = b.compact(full)
    line = json.dumps(c, separators=(",", ":"))
    assert len(line) < 4096, len(line)
    _check_contract(c)
    for k in ("roofline", "roofline_hbm", "roofline_greek", "roofline_solve"):
        assert set(c[k]) <= set(b.ROOF_KEYS) and "note" not in c[k] and c[k]["frac"] <= 1.0
    assert set(c["post_check"]) == {"pairs_checked", "bitwise_equal", "max_rel_diff"}
    for cid in ("3", "4", "5"):
        leg = c["other_configs"][cid]
        for k in ("value", "ms_per_step", "timed_region_s", "single_pair_ms", "dominant", "bitwise_equal"):
            assert k in leg, (cid, k)
    assert abs(c["value"] - full["value"]) <= 1e-4 * full["value"]
    assert json.loads(line) == c
