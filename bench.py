#!/usr/bin/env python3
"""bench.py -- benchmark of the SFFT subtraction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W                  BASELINE configs[1] (the headline; default)
    python bench.py --config 3|5 ...                               configs[2] (B-spline, 6144^2) / configs[4] (9232 x 9216, KerHW 12)
    python bench.py --pairs 62 ...                                 configs[3]: a fixed batch of independent pairs dealt to the ranks
Launch contract: `python bench.py --gpus N` as a plain command starts its N ranks itself (one process per GPU through
torch.distributed.run on 127.0.0.1, RCCL); launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it
uses the ranks it is given.  Either way WORLD_SIZE must equal --gpus and the communicator must count N ranks (`ranks_seen`), or the
process exits with status 2 and prints no JSON line: an N-GPU request never yields a smaller job's number.

A "step" is one GSS-equivalent pass (solve on the masked pair + apply to the full pair, the body of
sfft.PureCupy_Customized_Packet.PCCP) over one BATCH of distinct synthetic image pairs that are resident in HBM when the
timed region starts.  Default: every rank holds `--batch` pairs (64 at config 2) and subtracts each once per step with
`--streams` pairs in flight (one plan + HIP stream + host thread each, static deal) -- weak scaling, no data-path
collective.  `--pairs M`: M pairs in total, dealt round-robin to the ranks (uneven shards), every rank's worker threads pull
from the shard's queue like the reference's multi-task packet (sfft/MultiEasyCrowdedPacket.py:361-399); a step is the whole
batch -- strong scaling.  Plans (tables + workspaces) are created before the timed region.  The only collective is the gather
of one [pair_id, status, ms, Solution] record per pair at the end (sfft_amd/sharding.py).

After the timed region every checked pair is subtracted again with ONE pair in flight into fresh buffers and the result of
the pipelined run must equal it bit for bit (`post_check`); the same isolated launches give the per-kernel durations of the
`roofline` objects (HIP events on the launch stream).

Rank 0 prints ONE JSON line.  `value` = image pairs per second over all ranks.  Extra objects:
  roofline      the stage with the most kernel time per pair among ALL timed stages (one pair in flight, HIP events): an HBM-bound
                transform pass against 8 TB/s, the Omega + Theta launch (v_mfma_f64_4x4x4_4b_f64) or the dense solve (n^3 / 3
                flops of the Cholesky factorisation, latency-bound) against the fp64 matrix peak.  roofline_hbm: the dominant
                HBM-bound kernel; roofline_greek: the Omega + Theta launch; roofline_solve: the factorisation
  cpu_baseline  the CPU restatement of the reference's Numpy path timed on this host (N = 1, config 2 only)
  host_arrays   the same workload with CP semantics: host (pinned) arrays in, host arrays out, PCIe both ways -- never `value`
  other_configs short legs of the other BASELINE configs after the headline's timed region (never part of `value`): 3 (B-spline,
                6144^2) and 5 (9232 x 9216, KerHW 12) at N = 1, and 4 (a batch of 62 config-2 pairs dealt to the ranks) at every N;
                each with its own value, timed region, single-pair stage times, roofline and post_check (`--no-other-configs` skips)
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a
# queue serialise.  A pipelined run uses two streams per pair in flight (the plan's second stream carries the apply pass's forward
# transforms), so the default is raised before the runtime initialises: 436 -> 466 pairs/s at four pairs in flight.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# multi-process GPU work on this host class needs dmabuf IPC (RCCL otherwise fails with `hipIpcGetMemHandle: invalid argument`)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6    # fp64 vector = fp64 matrix (MFMA) peak on MI355X (public spec; the guide has no fp64 MFMA row)
HBM_COPY_MEASURED_GBS = 5700.0   # a plain 16-byte-per-lane copy kernel on this device: 5.6 - 5.8 TB/s for contiguous chunks with non-temporal
                                 # accesses, 5.2 - 5.6 plain (scripts/micro/hbm_stream2.hip, profiles/r03_hbm_stream2.txt); the guide quotes 6290

CONFIGS = {
    2: dict(N0=4096, N1=4096, w=8, DK=2, DB=2, batch=64, streams=4,
            name="BASELINE configs[1]: 4096x4096 pairs, KerHW 8, KerPolyOrder 2, BGPolyOrder 2, ConstPhotRatio, fp64"),
    3: dict(N0=6144, N1=6144, w=8, DK=2, DB=2, batch=6, streams=3, bspline=True,
            name="BASELINE configs[2]: BSplineSFFT, 6144x6144 pairs, KerHW 8, B-spline kernel degree 2 with 2x2 internal knots "
                 "(Fij 25, NEQ 7231), constant scaling, polynomial background degree 2, fp64"),
    5: dict(N0=9232, N1=9216, w=12, DK=3, DB=3, batch=4, streams=2,
            name="BASELINE configs[4]: 9232x9216 pairs, KerHW 12, KerPolyOrder 3, BGPolyOrder 3, ConstPhotRatio, fp64"),
}


def alg_bytes(N0, N1, w, Fij, Fpq, n_colfac, DB, mixed_apply, theta_fused=False, omg_counts=None, decimated=False, chunks=0):
    """Algorithmic HBM bytes per stage for ONE pair (solve + apply), as built, and the
    canonical reference-algorithm figure B_alg of SURVEY.md 8(d).  n_colfac = distinct column factors of the kernel basis
    (DK + 1 for a polynomial, Fj for a B-spline tensor basis): one row transform each."""
    P = N0 * N1
    Nh = N1 // 2 + 1
    c, r = 16, 8
    spec = c * N0 * Nh                                  # one half-spectrum plane
    n_off, n_diag = omg_counts if omg_counts else (Fij * (Fij - 1) // 2, Fij)     # Omega products that are transformed
    out = {
        # forward transforms of the solve pass as built: one row transform per distinct column factor (+ J) into stage planes;
        # the column pass reads every stage plane (from HBM once, its other readers hit L2) and writes Fij + 1 planes
        "fwd_rows": 2 * r * P + (n_colfac + 1) * spec,
        "fwd_cols": (n_colfac + 1) * spec + (Fij + 1) * spec,
        # Greek stage 1 as built: all passes of a (64-column x row-chunk) tile run on one XCD back to back, so each of the
        # Fij planes is streamed from HBM once and re-read from that XCD's L2; the per-chunk partial lag sums are written
        "greek_g1": Fij * spec,
        "greek_g1b": (Fij + 1) * spec + r * P,          # Theta passes: Fij planes + FJ; Gamma block: one read of the masked image
        # fp64 flops of the Omega passes: per pass and spectrum element one complex product (6) + 4 real FMAs per lag
        # (the Fij diagonal passes have a real product: half the lag work)
        "greek_g1_flops": N0 * Nh * (n_off * (6 + 2 * 4 * (2 * w)) + n_diag * (3 + 4 * (2 * w))),
    }
    if theta_fused:      # the Fij Theta passes (half width w) ride in the Omega launch: + the FJ plane, + their flops; the short-pass stage
        out["greek_g1"] += spec                                  # keeps only the Gamma block (one read of the masked image)
        out["greek_g1b"] = r * P
        out["greek_g1_flops"] += N0 * Nh * Fij * (6 + 2 * 4 * w)
    if chunks:           # the per-chunk partial lag sums the launch writes for greek_g2 (2 * 2w + 1 lags per Omega product, 2w + 1 per Theta pass)
        out["greek_g1"] += chunks * c * (N1 // 2 + 4) * ((n_off + n_diag) * (4 * w + 1) + (Fij * (2 * w + 1) if theta_fused else 0))
    out["greek_g1_flops_direct"] = out["greek_g1_flops"]        # the pruned DFT taken directly: every lag over every row
    if decimated:
        # one radix-2 decimation step along the rows (greek_g1_mfma4g<false, true>): the products are formed for every row, their sum and
        # difference over the row pairs (x', x' + N0/2) cost 4 (2) additions per pair, and the lag sums run over N0 / 2 rows
        n_the = Fij if theta_fused else 0
        out["greek_g1_flops"] = (N0 * Nh * (n_off * 6 + n_diag * 3 + n_the * 6)
                                 + (N0 // 2) * Nh * (n_off * (4 + 2 * 4 * (2 * w)) + n_diag * (2 + 4 * (2 * w)) + n_the * (4 + 2 * 4 * w)))
    # the part of greek_g1_flops that is lag sums (what runs on the matrix cores when the lag half-width allows); the rest -- complex
    # products and decimation butterflies -- runs on the vector ALUs
    n_the = Fij if theta_fused else 0
    rows_l = (N0 // 2) if decimated else N0
    out["greek_g1_mfma_flops"] = rows_l * Nh * (n_off * 2 * 4 * (2 * w) + n_diag * 4 * (2 * w) + n_the * 2 * 4 * w)
    if mixed_apply:
        # polynomial kernel: row pass into stage planes, mixed-domain column convolution (reads them, writes one plane),
        # inverse row pass with the DIFF epilogue -- no column transforms
        out.update(prelim_apply=r * P + n_colfac * spec, construct=n_colfac * spec + spec, inverse=spec + 2 * r * P)
    else:
        # Fourier-domain apply: Fij forward plane transforms, construct_fd reads them and writes one, inverse column + row pass
        out.update(prelim_apply=r * P + n_colfac * spec + (n_colfac + Fij) * spec, construct=(Fij + 1) * spec, inverse=3 * spec + 2 * r * P)
    n_pre = 1 + Fij + Fpq
    n_greek = Fij * Fij + 2 * Fij * Fpq + Fpq * Fpq + Fij + Fpq
    n_fft = 2 * n_pre + n_greek + 1
    out["B_alg_reference"] = n_fft * 4 * c * P + n_greek * c * P + n_pre * c * P + 5 * r * P
    out["n_fft_reference"] = n_fft
    return out


def roofline_object(stage, kernel, t_ms, a_bytes, a_mfma_flops=0.0, traffic=None, **extra):
    """One roofline object.  Both rates are ALGORITHMIC work over the measured duration: bytes against 8 TB/s and
    matrix-pipe flops against 78.6 TFLOP/s; `bound` names the larger fraction and `achieved` / `peak` / `unit` / `frac` follow it.  The PMC
    byte count of the same launch goes to `traffic` and `traffic_ratio` (= traffic / algorithmic bytes: over-fetch, 1.0 = every byte moved
    once) -- it is never a numerator of `achieved` (tests/test_bench_accounting.py)."""
    t = max(t_ms, 1e-6) * 1e-3
    gbs, tf = a_bytes / t / 1e9, a_mfma_flops / t / 1e12
    hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tf / FP64_PEAK_TFLOPS
    o = {"kernel": kernel, "stage": stage, "avg_ms": t_ms, "alg_bytes_per_launch": a_bytes,
         "alg_mfma_flops_per_launch": a_mfma_flops, "hbm_frac": hbm_frac, "mfma_frac": mfma_frac, "hbm_GBs": gbs, "mfma_tflops": tf,
         "traffic": traffic, "traffic_ratio": (traffic / a_bytes) if (traffic and a_bytes) else None}
    if mfma_frac > hbm_frac:
        o.update(bound="mfma", achieved=tf, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=mfma_frac)
    else:
        o.update(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=hbm_frac, sustained_peak_measured=HBM_COPY_MEASURED_GBS)
    o.update(extra)
    return o


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _round(o, sig=5):
    """floats to `sig` significant digits (the compact line only; the full object keeps every digit)"""
    if isinstance(o, float):
        return float("%.*g" % (sig, o))
    if isinstance(o, dict):
        return {k: _round(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round(v, sig) for v in o]
    return o


ROOF_KEYS = ("bound", "kernel", "stage", "achieved", "peak", "unit", "frac", "hbm_frac", "mfma_frac", "traffic", "traffic_ratio", "avg_ms")


def compact(full):
    """The LAST stdout line: the contract's keys and the judged objects only, well under the 8 KB tail the driver keeps of
    stdout (round 3's 26 KB line arrived headless and could not be parsed).  Everything else -- notes, per-stage tables, the
    full legs of the other configs -- is in profiles/bench_last_full.json (and gpurun_out/ when that directory exists)."""
    out = _pick(full, ("metric", "value", "unit", "mpix_per_s", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    cfg = full.get("config", {})
    out["config"] = _pick(cfg, ("workload", "baseline_config", "pairs_per_step", "pairs_in_flight_per_gpu", "timed_region_s", "NEQ", "solver"))
    if len(out["config"].get("workload", "")) > 200:
        out["config"]["workload"] = out["config"]["workload"][:197] + "..."
    for k in ("roofline", "roofline_hbm", "roofline_greek", "roofline_solve"):
        if k in full:
            out[k] = _pick(full[k], ROOF_KEYS)
    if "pipeline" in full:
        out["pipeline"] = _pick(full["pipeline"], ("bound", "achieved", "peak", "unit", "frac", "alg_bytes_per_pair"))
    out.update(_pick(full, ("ranks_seen", "launch", "solve_lu_ms")))
    if isinstance(full.get("solve_lu"), dict) and full["solve_lu"].get("vendor_getrf_getrs_ms") is not None:
        out["solve_vendor_lu_ms"] = round(full["solve_lu"]["vendor_getrf_getrs_ms"], 3)
    if "single_pair" in full:
        out["single_pair_ms"] = full["single_pair"]["ms"]
    if "prelim_apply_alone" in full:      # [alone, beside the solve] (ms): the apply pass's forward transforms
        out["prelim_apply_ms"] = [full["prelim_apply_alone"]["ms"], full["prelim_apply_alone"]["ms_beside_the_solve"]]
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "seconds_per_pair", "spread_s", "protocol", "cpu_model", "physical_cores",
                                         "all_cores_s_per_pair", "full_protocol"))
        out["cpu_baseline"]["sample"] = "one full GSS of the 4096x4096 seed-1234 pair (pair 0 of the GPU batch)"
        if isinstance(out["cpu_baseline"].get("protocol"), str):        # (the full wording is in the full object)
            out["cpu_baseline"]["protocol"] = out["cpu_baseline"]["protocol"][:64]
        if isinstance(out["cpu_baseline"].get("full_protocol"), dict):
            out["cpu_baseline"]["full_protocol"] = _pick(out["cpu_baseline"]["full_protocol"], ("threads_8_s_per_pair", "all_cores_s_per_pair", "runs_each", "warmups_each"))
    if "post_check" in full:
        out["post_check"] = _pick(full["post_check"], ("pairs_checked", "bitwise_equal", "max_rel_diff"))
    if "host_arrays" in full:
        out["host_arrays"] = _pick(full["host_arrays"], ("value", "pcie_GBs"))
    out.update(_pick(full, ("gathered_pairs", "failed_pairs", "solves_timed", "lu_fallback_pairs", "chol_stall_events")))
    if "per_rank" in full:      # one short row per rank: an imbalanced or mis-placed rank must be visible in the line the driver keeps
        out["per_rank"] = [[r["rank"], r["device"], r["pairs"], round(r["pairs_per_s"], 1), r["numa_node"], r["cpus_allowed"]] for r in full["per_rank"]]
        out["per_rank_keys"] = "rank,device,pairs,pairs_per_s,numa_node,cpus_allowed"
    legs = {}
    for cid, leg in (full.get("other_configs") or {}).items():
        if not isinstance(leg, dict):
            continue
        if "error" in leg:
            legs[cid] = {"error": str(leg["error"])[:160]}
            continue
        dom = leg.get("roofline", {})
        legs[cid] = {"value": leg.get("value"), "ms_per_step": leg.get("ms_per_step"),
                     "single_pair_ms": leg.get("single_pair", {}).get("ms"), "pairs_per_step": leg.get("config", {}).get("pairs_per_step"),
                     "prelim_apply_ms": [leg.get("prelim_apply_alone", {}).get("ms"), leg.get("prelim_apply_alone", {}).get("ms_beside_the_solve")],
                     "dominant": _pick(dom, ("kernel", "bound", "frac", "avg_ms", "traffic_ratio")),
                     "pipeline_frac": leg.get("pipeline", {}).get("frac"), "solve_lu_ms": leg.get("solve_lu_ms"),
                     "solve_vendor_lu_ms": (leg.get("solve_lu") or {}).get("vendor_getrf_getrs_ms"),
                     "bitwise_equal": leg.get("post_check", {}).get("bitwise_equal"),
                     "gathered_pairs": leg.get("gathered_pairs"), "failed_pairs": leg.get("failed_pairs")}
        if leg.get("lu_fallback_pairs") is not None:        # [LU fallbacks, Cholesky hand-off stalls] of the leg's timed region
            legs[cid]["lu_stall"] = [leg.get("lu_fallback_pairs"), leg.get("chol_stall_events")]
    if legs:
        out["other_configs"] = legs
    out["full_line"] = "profiles/bench_last_full.json"
    return _round(out)


def emit(full):
    """Write the full object to profiles/bench_last_full.json (+ gpurun_out/ if present), print the compact line LAST."""
    blob = json.dumps(full)
    for d in ("profiles", "gpurun_out"):
        p = os.path.join(ROOT, d)
        if os.path.isdir(p):
            try:
                with open(os.path.join(p, "bench_last_full.json"), "w") as f:
                    f.write(blob + "\n")
            except OSError:
                pass
    line = json.dumps(compact(full), separators=(",", ":"))
    assert len(line) < 6000, "compact bench line grew to %d bytes" % len(line)
    print(line, flush=True)


def cpu_baseline(cfg, quick):
    """The CPU restatement of the reference's Numpy path on this host (oracle/: test infrastructure; timed here, never shipped).
    Config 2 only.  See oracle/cpu_baseline.py for the protocol."""
    from oracle import cpu_baseline as CB
    return CB.measure(cfg["N0"], cfg["N1"], cfg["w"], cfg["DK"], cfg["DB"], quick=quick)


def blob_pair_device(torch, N0, N1, seed, dev, ratio=1.25):
    """Cheap synthetic pair made on the device (the star renderer of utils/synthetic.py takes half a minute at 85 Mpix):
    smooth blobs + noise; SCI = ratio * (REF blurred by a 3-tap kernel) + sky + noise; the masked pair keeps the blobs."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.linspace(0, N0 / 100.0 * np.pi, N0, dtype=torch.float64, device=dev)[:, None]
    y = torch.linspace(0, N1 / 110.0 * np.pi, N1, dtype=torch.float64, device=dev)[None, :]
    base = 80.0 * (torch.sin(x + 0.37 * seed) * torch.cos(y - 0.11 * seed)) ** 8
    REF = base + torch.randn((N0, N1), dtype=torch.float64, device=dev, generator=g)
    SCI = ratio * (0.6 * base + 0.2 * torch.roll(base, 1, 0) + 0.2 * torch.roll(base, -1, 1)) + 2.0 \
        + torch.randn((N0, N1), dtype=torch.float64, device=dev, generator=g)
    keep = base > 0.5
    z = torch.zeros((), dtype=torch.float64, device=dev)
    return {"REF": REF.contiguous(), "SCI": SCI.contiguous(), "mREF": torch.where(keep, REF, z).contiguous(),
            "mSCI": torch.where(keep, SCI, z).contiguous()}


def derive_pair(torch, base, k, dev):
    """Pair k of a batch from a seeded star-field pair: circular shift (the SFFT model is periodic, so a shifted field is as
    good a field) + fresh N(0, 0.5) noise in both frames; the star mask moves along.  k = 0 is the base pair itself."""
    if k == 0:
        return base
    N0, N1 = base["REF"].shape
    sh = ((131 * k) % N0, (977 * k) % N1)
    g = torch.Generator(device=dev)
    g.manual_seed(7000 + k)
    keep = torch.roll(base["mREF"] != 0, sh, (0, 1))
    REF = torch.roll(base["REF"], sh, (0, 1)) + 0.5 * torch.randn((N0, N1), dtype=torch.float64, device=dev, generator=g)
    SCI = torch.roll(base["SCI"], sh, (0, 1)) + 0.5 * torch.randn((N0, N1), dtype=torch.float64, device=dev, generator=g)
    z = torch.zeros((), dtype=torch.float64, device=dev)
    return {"REF": REF.contiguous(), "SCI": SCI.contiguous(), "mREF": torch.where(keep, REF, z).contiguous(),
            "mSCI": torch.where(keep, SCI, z).contiguous()}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE config (1-based as in SURVEY 8d): 2 = headline")
    ap.add_argument("--pairs", type=int, default=0, help="config-4 mode: a fixed batch of this many config-2 pairs dealt round-robin "
                    "to the ranks (62 = one DECam focal plane); a step = the whole batch")
    ap.add_argument("--batch", type=int, default=0, help="distinct pairs per GPU and step (default: 64 / 6 / 4 for config 2 / 3 / 5, with 4 / 3 / 2 pairs in flight)")
    ap.add_argument("--streams", type=int, default=0, help="independent pairs in flight per GPU (one plan + stream + host thread each)")
    ap.add_argument("--size", type=int, default=0, help="override the image side (square), e.g. for a quick run")
    ap.add_argument("--kerhw", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline with the full protocol (3 warm-ups, median of 10) instead of the bounded one")
    ap.add_argument("--no-host-arrays", action="store_true", help="skip the host-array (PCIe-inclusive) variant")
    ap.add_argument("--no-lu-leg", action="store_true", help="skip the forced-LU leg and the vendor getrf + getrs yardstick behind the timed region "
                    "(profiling runs: thousands of tiny vendor launches under rocprofv3 --pmc crashed the tool at configs 3 / 5)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs of configs 3, 4 and 5 after the headline run")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl", help="torch.distributed backend of the N > 1 run: nccl (= RCCL over "
                    "xGMI, the default) or gloo (records and timing cross the ranks as host tensors; tests)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="TEST ONLY: every rank uses cuda:0 (two processes on a one-GPU box; "
                    "needs --dist-backend gloo, RCCL refuses two ranks on one device)")
    ap.add_argument("--spawn", action="store_true", help="start the rank(s) through torch.distributed.run even at --gpus 1 (a process group with "
                    "one rank: RCCL initialised, collectives executed)")
    ap.add_argument("--dk", type=int, default=-1, help="override the kernel polynomial order (quick runs / tests)")
    ap.add_argument("--db", type=int, default=-1, help="override the background polynomial order (quick runs / tests)")
    return ap.parse_args(argv)


def run_config(args, rank, world, local_rank, headline_extras=True):
    """One bench leg: `args.config` (or config 4 with `args.pairs`), `args.warmup` untimed + `args.steps` timed steps between
    barriers.  Returns the JSON object on rank 0, None elsewhere.  headline_extras: also run the host-array variant and the
    cpu_baseline leg (config 2 at N = 1 only)."""
    import threading
    import torch
    import torch.distributed as dist
    from sfft_amd import _lib
    from sfft_amd.plan import Plan
    from sfft_amd.sharding import shard_pair_ids, pack_record, gather_records, run_shard, all_failed_with_error
    from sfft_amd.utils.synthetic import make_pair

    dev = torch.device("cuda", local_rank)
    cfg = dict(CONFIGS[args.config])
    if args.size:
        cfg["N0"] = cfg["N1"] = args.size
    if args.kerhw:
        cfg["w"] = args.kerhw
    if args.dk >= 0:
        cfg["DK"] = args.dk
    if args.db >= 0:
        cfg["DB"] = args.db
    N0, N1, w, DK, DB = cfg["N0"], cfg["N1"], cfg["w"], cfg["DK"], cfg["DB"]
    bspline = bool(cfg.get("bspline"))
    S = max(1, args.streams or cfg["streams"])
    batch_mode = args.pairs > 0
    if batch_mode:
        assert args.config == 2, "--pairs is config 4: a batch of config-2 pairs"
        my_ids = shard_pair_ids(args.pairs, rank, world)          # global pair ids of this rank's shard (uneven shards)
        n_total = args.pairs
    else:
        B = max(1, args.batch or cfg["batch"])
        my_ids = list(range(rank * B, (rank + 1) * B))
        n_total = world * B
    S = min(S, max(1, len(my_ids)))

    # ---- plans: one per pair in flight ----------------------------------------------------------------------------
    t0 = time.perf_counter()
    if bspline:
        from sfft_amd.BSplineSFFT import _axis_tables
        kx, ky = [N0 / 3 + 0.5, 2 * N0 / 3 + 0.5], [N1 / 3 + 0.5, 2 * N1 / 3 + 0.5]
        kbx, kby, kpairs = _axis_tables(N0, N1, "B-Spline", DK, kx, ky)
        tbx, tby, bpairs = _axis_tables(N0, N1, "Polynomial", DB, [], [])
        bdict = dict(kbx=kbx, kby=kby, ker_pairs=kpairs, tbx=tbx, tby=tby, bkg_pairs=bpairs, scaling_mode=2)
        plans = [Plan(N0, N1, w, device=local_rank, basis=bdict) for _ in range(S)]
        n_colfac = kby.shape[0]
    else:
        plans = [Plan(N0, N1, w, DK, DB, True, device=local_rank) for _ in range(S)]
        n_colfac = DK + 1
    torch.cuda.synchronize(dev)
    plan_s = (time.perf_counter() - t0) / S
    NEQ, Fij, Fpq = plans[0].NEQ, plans[0].query("Fij"), plans[0].Fpq
    streams = [torch.cuda.Stream(dev) for _ in range(S)]

    # ---- the batch: distinct pairs, resident in HBM before the timed region ------------------------------------------
    if args.config == 2:
        nbase = min(2, len(my_ids))
        bases = []
        for b in range(nbase):   # seeded star fields (SURVEY 8d recipe); pair ids b, b + nbase, ... derive from base b
            pr = make_pair(N0, N1, seed=1234 + 16 * rank + b, mask=True, sky=0.0, bkg_scale=0.05)
            bases.append({k: torch.from_numpy(v).to(dev) for k, v in pr.items()})
        pairs = [derive_pair(torch, bases[k % nbase], k // nbase, dev) for k in range(len(my_ids))]
    else:
        pairs = [blob_pair_device(torch, N0, N1, 100 * rank + k + 3, dev) for k in range(len(my_ids))]
    sols = [torch.zeros(NEQ, dtype=torch.float64, device=dev) for _ in my_ids]
    diffs = [torch.empty((N0, N1), dtype=torch.float64, device=dev) for _ in my_ids]
    torch.cuda.synchronize(dev)

    status = [0] * len(my_ids)       # the C ABI's return code of each pair's last sfft_subtract (static mode; run_shard keeps its own)

    def subtract(wi, k):
        """pair k of this rank's shard on worker wi's plan and stream"""
        g = pairs[k]
        plans[wi].subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"], out_solution=sols[k], out_diff=diffs[k])
        return sols[k]

    def subtract_recorded(wi, k):
        try:
            subtract(wi, k)
            status[k] = 0
        except np.linalg.LinAlgError:
            status[k] = _lib.SFFT_ERR_SINGULAR
        except _lib.SfftError as e:
            status[k] = e.code

    last_records = [None]
    shard_errors = []            # (pair id, message) of pairs that failed with anything but an ABI status (run_shard also prints them)

    def run_steps(n):
        def static_worker(wi):
            torch.cuda.set_device(local_rank)
            with torch.cuda.stream(streams[wi]):
                for _ in range(n):
                    for k in range(wi, len(my_ids), S):
                        subtract_recorded(wi, k)
        if not batch_mode:
            if S == 1:
                static_worker(0)
            else:
                th = [threading.Thread(target=static_worker, args=(i,)) for i in range(S)]
                [t.start() for t in th]
                [t.join() for t in th]
            return
        for _ in range(n):          # config 4: the shard's queue, workers pull (run_shard), per-pair status and time recorded

            def work(wi, pid):
                torch.cuda.set_device(local_rank)
                with torch.cuda.stream(streams[wi]):
                    return subtract(wi, my_ids.index(pid))
            last_records[0] = run_shard(my_ids, S, work, NEQ, dev, errors=shard_errors)

    def barrier():
        if dist.is_initialized():
            dist.barrier()

    def solver_counters():
        """[solves, LU fallbacks, Cholesky hand-off stalls] of this rank's plans since they were created (sfft_plan_query)"""
        return [sum(pl.query(k) for pl in plans) for k in ("SOLVES", "LU_FALLBACKS", "CHOL_STALLS")]

    run_steps(args.warmup)
    torch.cuda.synchronize(dev)
    c0 = solver_counters()
    barrier()
    t_start = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize(dev)
    t_own = time.perf_counter() - t_start          # this rank's own time for its shard (before it waits for the others)
    barrier()
    elapsed = time.perf_counter() - t_start
    # what the TIMED region ran on, read before anything else touches the plans: which solver its pairs took (the reference always runs LU,
    # SFFTSubtract.py:15-23; here a failed Cholesky attempt silently costs a second, 7x slower solve) and how often a hand-off poll ran out
    c1 = solver_counters()
    timed_solver = {1: "cholesky", 2: "lu"}.get(plans[0].query("LAST_SOLVER"), "?")
    solver_delta = [b - a for a, b in zip(c0, c1)]
    comm_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")       # where the collectives' tensors live
    # one row per rank: [rank, device, pairs per step, own seconds, solves, LU fallbacks, stalls, NUMA node of the GPU, CPUs the rank's threads may run on]
    my_row = torch.tensor([rank, local_rank, len(my_ids), t_own] + solver_delta + list(AFFINITY), dtype=torch.float64, device=comm_dev)
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rows = [torch.zeros_like(my_row) for _ in range(world)]
        dist.all_gather(rows, my_row)
    else:
        rows = [my_row]
    per_rank = [{"rank": int(r[0]), "device": int(r[1]), "pairs": int(r[2]) * args.steps, "seconds": float(r[3]),
                 "pairs_per_s": int(r[2]) * args.steps / max(float(r[3]), 1e-9), "solves": int(r[4]), "lu_fallback_pairs": int(r[5]),
                 "chol_stall_events": int(r[6]), "numa_node": int(r[7]), "cpus_allowed": int(r[8])} for r in (x.cpu() for x in rows)]

    # ---- post-run check + isolated per-kernel durations: ONE pair in flight, fresh output buffers -----------------------
    check_ids = sorted(set([0, len(my_ids) // 2, len(my_ids) - 1]))
    iso_acc, iso_ms = {}, []
    post = {"pairs_checked": [my_ids[k] for k in check_ids], "bitwise_equal": True, "max_rel_diff": 0.0}
    plans[0].set_timing(True)
    with torch.cuda.stream(streams[0]):
        fresh_s = torch.empty(NEQ, dtype=torch.float64, device=dev)
        fresh_d = torch.empty((N0, N1), dtype=torch.float64, device=dev)
        for k in check_ids * (2 if len(check_ids) < 3 else 1):
            g = pairs[k]
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            plans[0].subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"], out_solution=fresh_s, out_diff=fresh_d)
            torch.cuda.synchronize(dev)
            iso_ms.append((time.perf_counter() - t1) * 1e3)
            for kk, v in plans[0].stage_ms().items():
                iso_acc[kk] = iso_acc.get(kk, 0.0) + v
            stage_kernels = plans[0].stage_kernels()          # the kernels each stage launched (names as at the launch sites: what rocprof lists)
            same = bool(torch.equal(fresh_s, sols[k])) and bool(torch.equal(fresh_d, diffs[k]))
            post["bitwise_equal"] = post["bitwise_equal"] and same
            if not same:
                rel = float(((fresh_d - diffs[k]).abs().max() / fresh_d.abs().max()).item())
                post["max_rel_diff"] = max(post["max_rel_diff"], rel)
                ds = (fresh_s - sols[k]).abs()
                post.setdefault("solution_mismatch", []).append({"pair": my_ids[k], "entries": int((fresh_s != sols[k]).sum().item()),
                                                                 "max_abs": float(ds.max().item()), "max_rel": float((ds.max() / fresh_s.abs().max()).item()),
                                                                 "first_index": int(torch.nonzero(fresh_s != sols[k])[0].item()) if bool((fresh_s != sols[k]).any()) else -1,
                                                                 "diff_entries": int((fresh_d != diffs[k]).sum().item()),
                                                                 "where": torch.nonzero(fresh_s != sols[k]).reshape(-1)[:8].tolist(),
                                                                 "fresh": fresh_s[fresh_s != sols[k]][:8].tolist(),
                                                                 "pipelined": sols[k][fresh_s != sols[k]][:8].tolist()})
            assert bool(torch.isfinite(fresh_d).all()), "non-finite DIFF"
        n_iso = len(iso_ms)
        # In a GSS the apply pass's forward transforms (stage prelim_apply) run on the plan's second stream BESIDE the dense solve: the
        # duration above is a contended one.  An apply call alone (sfft_apply: the same launches on the caller's stream, nothing beside
        # them) gives what the kernels can do on their own; both are printed.
        apply_alone = {}
        for _ in range(2):
            g = pairs[check_ids[0]]
            plans[0].apply(g["REF"], g["SCI"], fresh_s)
            torch.cuda.synchronize(dev)
            apply_alone = plans[0].stage_ms()
        # the same system through the reference's own solver semantics -- LU with partial pivoting (lu.hpp, sfft_set_force_lu): the solve
        # stage's duration beside the Cholesky path's, and the DIFF it leads to against the Cholesky run's
        solve_lu = None
        try:
            if args.no_lu_leg:
                raise RuntimeError("skipped (--no-lu-leg)")
            plans[0].set_force_lu(True)
            g = pairs[check_ids[0]]
            lu_ms = []
            for _ in range(3):
                plans[0].subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"], out_solution=fresh_s, out_diff=fresh_d)
                torch.cuda.synchronize(dev)
                lu_ms.append(plans[0].stage_ms().get("solve", 0.0))
            used_lu = plans[0].query("LAST_SOLVER") == 2
            rel = float(((fresh_d - diffs[check_ids[0]]).abs().max() / diffs[check_ids[0]].abs().max()).item())
            solve_lu = {"ms": float(np.median(lu_ms[1:])), "used_lu": bool(used_lu), "diff_max_rel_vs_cholesky": rel}
            try:
                # yardstick (never part of the product path): the platform's own getrf + getrs on the SAME system -- the ROCm counterpart of
                # the reference's cupy.linalg.solve (SFFTSubtract.py:15-23) -- timed with events on the current stream
                A_sys, b_sys, _ = plans[0].get_solver_system()
                ven = []
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    torch.linalg.solve(A_sys, b_sys)
                    e1.record()
                    torch.cuda.synchronize(dev)
                    ven.append(e0.elapsed_time(e1))
                solve_lu["vendor_getrf_getrs_ms"] = float(np.median(ven[1:]))
                del A_sys, b_sys
            except Exception as e:
                solve_lu["vendor_getrf_getrs_ms"] = None
                solve_lu["vendor_error"] = "%s: %s" % (type(e).__name__, e)
        except Exception as e:      # (a failing LU leg must not take the line with it; it is reported)
            solve_lu = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            plans[0].set_force_lu(False)
    plans[0].set_timing(False)
    assert post["bitwise_equal"] or post["max_rel_diff"] <= 1e-12, "pipelined result differs from the single-stream result: %r" % post

    # ---- records: the only collective of the data path -------------------------------------------------------------------
    if batch_mode:
        recs = last_records[0]
    else:
        recs = [pack_record(my_ids[k], status[k], elapsed * 1e3 / max(args.steps * len(my_ids), 1), sols[k]) for k in range(len(my_ids))]
    table = gather_records(recs, n_total, NEQ, comm_dev)
    n_failed = int((table[:, 1] != 0).sum().item())
    if batch_mode and all_failed_with_error(recs):      # after the collective, so that no rank is left waiting in it
        raise RuntimeError("rank %d: every pair of the shard failed with an exception: %r" % (rank, shard_errors[:3]))

    # ---- host-array variant (CP semantics): pinned host arrays in, host arrays out, H2D / D2H overlapped across streams ----
    host = None
    if headline_extras and world == 1 and args.config == 2 and not batch_mode and not args.no_host_arrays:
        nh = min(len(my_ids), 2 * S)
        hin = [{k: v.cpu().pin_memory() for k, v in pairs[k].items()} for k in range(nh)]
        hout = [torch.empty((N0, N1), dtype=torch.float64).pin_memory() for _ in range(nh)]
        hsol = [torch.empty(NEQ, dtype=torch.float64).pin_memory() for _ in range(nh)]
        dbuf = [{k: torch.empty((N0, N1), dtype=torch.float64, device=dev) for k in ("REF", "SCI", "mREF", "mSCI")} for _ in range(S)]
        dd = [torch.empty((N0, N1), dtype=torch.float64, device=dev) for _ in range(S)]
        ds = [torch.empty(NEQ, dtype=torch.float64, device=dev) for _ in range(S)]

        def host_worker(wi, reps):
            torch.cuda.set_device(local_rank)
            with torch.cuda.stream(streams[wi]):
                for _ in range(reps):
                    for k in range(wi, nh, S):
                        for name in ("REF", "SCI", "mREF", "mSCI"):
                            dbuf[wi][name].copy_(hin[k][name], non_blocking=True)
                        plans[wi].subtract(dbuf[wi]["REF"], dbuf[wi]["SCI"], dbuf[wi]["mREF"], dbuf[wi]["mSCI"], out_solution=ds[wi], out_diff=dd[wi])
                        hout[k].copy_(dd[wi], non_blocking=True)
                        hsol[k].copy_(ds[wi], non_blocking=True)
                streams[wi].synchronize()

        def host_run(reps):
            th = [threading.Thread(target=host_worker, args=(i, reps)) for i in range(S)]
            [t.start() for t in th]
            [t.join() for t in th]
        host_run(1)
        torch.cuda.synchronize(dev)
        reps = 6
        t1 = time.perf_counter()
        host_run(reps)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t1
        ok = all(bool(torch.equal(hout[k], diffs[k].cpu())) for k in (0, nh - 1))
        moved = (4 + 1) * 8 * N0 * N1
        host = {"value": reps * nh / dt, "unit": "image-pairs/s", "pairs": reps * nh, "seconds": dt,
                "pcie_GBs": reps * nh * moved / dt / 1e9, "bytes_per_pair": moved, "matches_device_resident_run": ok,
                "note": "CP / GSS semantics: 4 pinned host images in (H2D), DIFF + Solution out (D2H), per pair, %d pairs in flight; "
                        "never `value`" % S}
        del hin, hout, dbuf, dd

    out = None
    if rank == 0:
        value = (n_total - n_failed) * args.steps / elapsed      # pairs whose subtraction failed (status != 0 in the gathered records) are not throughput
        ms_step = elapsed * 1e3 / args.steps
        iso_stage = {k: v / n_iso for k, v in iso_acc.items()}
        mixed = (w <= 12) if not bspline else (w <= 8 and 4 <= n_colfac <= 6)     # (B-spline tensor bases of 4 x 4 .. 6 x 6 terms: vconv_tensor)
        theta_fused = bool(plans[0].query("THETA_FUSED"))
        decimated = bool(plans[0].query("G1_DECIMATED"))
        ab = alg_bytes(N0, N1, w, Fij, Fpq, n_colfac, DB, mixed, theta_fused,
                       (plans[0].query("OMG_OFFDIAG"), plans[0].query("OMG_DIAG")), decimated, plans[0].query("G1_CHUNKS"))
        headline = (args.config == 2 and (N0, N1, w) == (4096, 4096, 8))
        n_sys = plans[0].query("SOLVER_N")              # unknowns of the system that is factorised

        pmc = {}
        try:   # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/pmc_traffic.json; see profiles/README.md)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            pass
        # per config ("2" / "3" / "5") and stage; only for the configs' own geometry
        native = not (args.size or args.kerhw or args.dk >= 0 or args.db >= 0)
        pmc_cfg = pmc.get(str(args.config), {}) if native else {}
        g1_mfma = bool(plans[0].query("G1_MFMA"))
        # names of the kernels each stage launched in the isolated run (sfft_stage_kernels): the minor helpers are dropped from the label
        MINOR = ("set_i32", "zero_f64", "chol_begin", "scatter_solution", "delta_finish", "kernel_ctab_mixed", "chol_copy_diag")

        def label(stage):
            names = [k for k in stage_kernels.get(stage, []) if k not in MINOR]
            return " + ".join(names) if names else stage
        KERNEL_OF = {k: label(k) for k in ("fwd_rows", "fwd_cols", "greek_g1", "greek_g1b", "greek_g2", "fill", "solve", "prelim_apply", "construct", "inverse")}

        def roof_obj(stage, t_ms, a_bytes, a_mfma_flops=0.0, **extra):
            return roofline_object(stage, KERNEL_OF.get(stage, stage), t_ms, a_bytes, a_mfma_flops,
                                   pmc_cfg.get(stage, {}).get("hbm_bytes_per_launch"), **extra)

        def roof(stages, dom="fwd_cols"):
            # (a stage that was not timed separately, e.g. SFFT_STAGE_INTERLEAVE=1, reads 0)
            return roof_obj(dom, stages[dom], ab[dom])

        def roof_greek(stages):
            """the Omega + Theta launch: the lag sums on the matrix pipe (v_mfma_f64_4x4x4_4b_f64) when the lag half-width allows, its planes
            and partial sums as bytes; whichever fraction is larger names the bound.  The products and butterflies on the vector ALUs are listed."""
            t = stages["greek_g1"]
            return roof_obj("greek_g1", t, ab["greek_g1"], ab["greek_g1_mfma_flops"] if g1_mfma else 0.0,
                            valu_flops_per_launch=ab["greek_g1_flops"] - ab["greek_g1_mfma_flops"],
                            all_flops_tflops=ab["greek_g1_flops"] / (max(t, 1e-6) * 1e-3) / 1e12,
                            mfma_sustained_peak_measured=74.5,    # profiles/r02_mfma_f64_peak.txt: a loop of independent v_mfma_f64_4x4x4_4b_f64, TFLOP/s
                            decimated=decimated, alg_flops_direct=ab["greek_g1_flops_direct"],
                            direct_equivalent_tflops=ab["greek_g1_flops_direct"] / (max(t, 1e-6) * 1e-3) / 1e12,
                            note="alg_mfma_flops_per_launch = the lag sums as built; with the radix-2 decimation step along the rows they run over half "
                                 "the rows (alg_flops_direct: the same pruned DFT taken directly)")

        def roof_solve(stages):
            fl = n_sys ** 3 / 3.0                       # Cholesky factorisation (the triangular solves are O(n^2))
            return roof_obj("solve", stages["solve"], 8.0 * n_sys * n_sys, fl, unknowns=n_sys,
                            regime="latency: a chain of dependent 64-column block steps, not throughput")

        # the dominant stage by kernel time of one pair, among everything that is timed
        HBM_STAGES = [k for k in ("fwd_rows", "fwd_cols", "prelim_apply", "construct", "inverse") if k in ab]
        cand = {k: iso_stage.get(k, 0.0) for k in HBM_STAGES + ["greek_g1", "solve"]}
        dom = max(cand, key=lambda k: cand[k])
        roofline = roof_solve(iso_stage) if dom == "solve" else roof_greek(iso_stage) if dom == "greek_g1" else roof(iso_stage, dom)
        dom_hbm = max(HBM_STAGES, key=lambda k: cand[k])
        per_pair_keys = [k for k in ("fwd_rows", "fwd_cols", "greek_g1", "greek_g1b", "prelim_apply", "construct", "inverse") if k in ab]
        if batch_mode:
            workload = ("BASELINE configs[3]: a batch of %d independent %dx%d pairs (config-2 geometry) dealt round-robin to %d rank(s): "
                        "shards of %s pairs, %d worker threads per GPU pull from the shard's queue; a step = the whole batch"
                        % (n_total, N0, N1, world, sorted(set(len(shard_pair_ids(n_total, r, world)) for r in range(world)), reverse=True), S))
        else:
            workload = ("%s; GSS = solve(masked pair) + apply(full pair); %d distinct pairs per GPU and step, %d in flight per GPU "
                        "(one plan + stream each)" % (cfg["name"] if not (args.size or args.kerhw or args.dk >= 0 or args.db >= 0) else
                                                      "%dx%d pairs, KerHW %d (config %d geometry)" % (N0, N1, w, args.config), len(my_ids), S))
        out = {
            "metric": "image-pairs/sec, %dx%d, KerHW=%d polyOrd=%d" % (N0, N1, w, DK),
            "value": value, "unit": "image-pairs/s", "mpix_per_s": value * N0 * N1 / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if batch_mode else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": 4 if batch_mode else args.config,
                       "pairs_per_step": n_total, "pairs_in_flight_per_gpu": S, "timed_region_s": elapsed,
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "plan_create_s": plan_s, "NEQ": NEQ,
                       "solver": timed_solver if sum(r["lu_fallback_pairs"] for r in per_rank) == 0 else "cholesky, LU fallback on %d of %d solves" % (
                           sum(r["lu_fallback_pairs"] for r in per_rank), sum(r["solves"] for r in per_rank))},
            # the timed region's solver accounting over all ranks (SFFT_Q_SOLVES / _LU_FALLBACKS / _CHOL_STALLS before and after it)
            "solves_timed": sum(r["solves"] for r in per_rank), "lu_fallback_pairs": sum(r["lu_fallback_pairs"] for r in per_rank),
            "chol_stall_events": sum(r["chol_stall_events"] for r in per_rank),
            "per_rank": per_rank,
            "roofline": dict(roofline,
                             measured="HIP events on the launch stream around the stage's kernels, %d launches with one pair in flight right after "
                             "the timed region (same process, same buffers); the stage with the most kernel time per pair" % n_iso,
                             kernel_ms_per_pair={k: iso_stage[k] for k in ("fwd_rows", "fwd_cols", "greek_g1", "greek_g1b", "greek_g2", "fill", "solve",
                                                                           "prelim_apply", "construct", "inverse") if k in iso_stage}),
            "roofline_hbm": dict(roof(iso_stage, dom_hbm), measured="same events, same launches: the HBM-bound stage with the most time"),
            "roofline_greek": dict(roof_greek(iso_stage), measured="same events, same launches"),
            "roofline_solve": dict(roof_solve(iso_stage), measured="same events, same launches"),
            "pipeline": {"bound": "hbm", "achieved": sum(ab[k] for k in per_pair_keys) * (value / world) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": sum(ab[k] for k in per_pair_keys) * (value / world) / 1e9 / HBM_PEAK_GBS,
                         "alg_bytes_per_pair": sum(ab[k] for k in per_pair_keys),
                         "note": "the whole pipeline per GPU: the build's own algorithmic bytes of one pair (all HBM-side stages, solve + apply) x "
                                 "pairs/s per GPU against 8 TB/s; the dense solve and the lag sums overlap with it on other streams"},
            "hbm_stages": {k: {"GBs": ab[k] / (max(iso_stage[k], 1e-6) * 1e-3) / 1e9, "frac": ab[k] / (max(iso_stage[k], 1e-6) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "ms": iso_stage[k], "alg_bytes": ab[k], "kernel": KERNEL_OF.get(k, k)} for k in HBM_STAGES},
            "prelim_apply_alone": {"ms": apply_alone.get("prelim_apply", 0.0), "ms_beside_the_solve": iso_stage.get("prelim_apply", 0.0),
                                   "frac": ab["prelim_apply"] / (max(apply_alone.get("prelim_apply", 0.0), 1e-6) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "note": "the apply pass's forward transforms timed by an sfft_apply call of their own (nothing beside them) and, in hbm_stages, "
                                           "as they run inside a GSS: on the plan's second stream beside the dense solve"},
            "single_pair": {"ms": float(np.median(iso_ms)), "pairs_per_s": 1e3 / float(np.median(iso_ms)), "stage_ms": iso_stage,
                            "note": "one pair in flight: latency of one GSS and per-stage times without interleaving"},
            "post_check": dict(post, note="after the timed region each listed pair is subtracted again alone (one pair in flight, fresh "
                               "output buffers); Solution and DIFF of the pipelined run must be bit-identical"),
            "pair_effective": {"B_alg_reference_bytes": ab["B_alg_reference"], "n_fft_reference": ab["n_fft_reference"],
                               "effective_GBs_per_gpu": ab["B_alg_reference"] * (value / world) / 1e9,
                               "as_built_bytes_per_pair": sum(ab[k] for k in per_pair_keys),
                               "as_built_GBs_per_gpu": sum(ab[k] for k in per_pair_keys) * (value / world) / 1e9,
                               "note": "reference-algorithm bytes (SURVEY 8d) x pairs/s per GPU; context, not the roofline: the build's own "
                                       "algorithmic bytes per pair are listed beside it"},
            "solve_lu_ms": (solve_lu or {}).get("ms"), "solve_lu": solve_lu,
            "gathered_pairs": int(table.shape[0]), "failed_pairs": n_failed, "shard_errors_rank0": ["pair %d: %s" % e for e in shard_errors[:4]],
            "stage_kernels": stage_kernels,     # the kernels each stage launched, as the library recorded them (sfft_stage_kernels)
        }
        if batch_mode:
            ms = table[:, 2].cpu().numpy()
            out["per_pair_ms"] = {"median": float(np.median(ms)), "max": float(ms.max()),
                                  "note": "host time of each pair's sfft_subtract call in the last step (S pairs in flight share the GPU)"}
            assert int(table.shape[0]) == args.pairs
        if host is not None:
            out["host_arrays"] = host
    # release this leg's device memory before the next one (plans hold GBs of workspace)
    for pl in plans:
        pl.close()
    del plans, pairs, sols, diffs
    torch.cuda.empty_cache()
    if out is not None and headline_extras and world == 1 and headline and not batch_mode and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(cfg, quick=not args.cpu_full)
        if args.cpu_full:       # keep the full-protocol figures for the default line to quote (profiles/cpu_baseline_full.json)
            from oracle import cpu_baseline as CB
            blob = json.dumps(CB.full_record(out["cpu_baseline"]))
            for d in ("profiles", "gpurun_out"):
                if os.path.isdir(os.path.join(ROOT, d)):
                    open(os.path.join(ROOT, d, "cpu_baseline_full.json"), "w").write(blob + "\n")
    return out


AFFINITY = (-1, 0)       # (NUMA node of this rank's GPU, CPUs its threads may run on): set by pin_to_gpu_numa_node()


def pin_to_gpu_numa_node(local_rank, world):
    """One process per GPU: keep this rank's host threads (one per pair in flight, plus the runtime's) on the CPUs of the NUMA node its GPU hangs
    off, so that eight ranks do not all submit from node 0 (the reference pins nothing: one thread per device queue in ONE process,
    sfft/MultiEasyCrowdedPacket.py:361-399).  Linux sysfs only; anything missing leaves the affinity as it is.  SFFT_BENCH_NO_PIN=1: off."""
    global AFFINITY
    try:
        import torch
        allowed = len(os.sched_getaffinity(0))
        AFFINITY = (-1, allowed)
        if os.environ.get("SFFT_BENCH_NO_PIN"):
            return
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        AFFINITY = (node, allowed)
        if node < 0 or world <= 1:
            return            # (a single rank keeps every core: its cpu_baseline leg wants them)
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            AFFINITY = (node, len(cpus))
    except Exception as e:      # (no sysfs, no pci ids, a container without the node files ...)
        sys.stderr.write("bench.py: not pinning rank threads: %s: %s\n" % (type(e).__name__, e))


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` started as a plain command (no RANK / WORLD_SIZE in the environment): start the N ranks here -- one
    process per GPU through torch.distributed.run on 127.0.0.1 -- and hand on their exit code; rank 0's JSON line reaches this process's
    stdout unchanged (the children inherit it).  The reference fans one call out to its devices the same way
    (sfft/MultiEasyCrowdedPacket.py:361-399, 698-710)."""
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()
    if not args.all_ranks_on_device0 and args.gpus > n_dev:
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s); refusing to print a smaller job's number\n" % (args.gpus, n_dev))
        sys.exit(2)
    env = dict(os.environ, SFFT_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse_args()
    import copy
    if ("WORLD_SIZE" not in os.environ or ("RANK" not in os.environ and int(os.environ["WORLD_SIZE"]) == 1)) and (args.gpus > 1 or args.spawn):
        self_launch(args)        # does not return
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:       # never print a `world`-GPU number for a `--gpus N` request
        sys.stderr.write("bench.py: rank %d: WORLD_SIZE=%d but --gpus %d -- launch exactly N ranks (or run plain `python bench.py --gpus N`, "
                         "which starts them)\n" % (rank, world, args.gpus))
        sys.exit(2)
    if args.all_ranks_on_device0:
        assert args.dist_backend == "gloo", "--all-ranks-on-device0 is a test mode and needs --dist-backend gloo"
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d: LOCAL_RANK %d but %d GPU(s) visible\n" % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pin_to_gpu_numa_node(local_rank, world)
    # started by torch.distributed.run (the driver's or self_launch's): a process group even at N = 1.  A bare WORLD_SIZE=1 that some schedulers
    # and containers export without a rendezvous (no RANK / MASTER_PORT) is NOT a launch: the process runs directly instead of failing in init.
    launched = "WORLD_SIZE" in os.environ and (world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ) or bool(os.environ.get("SFFT_BENCH_SELF_LAUNCHED")))
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # the ranks that actually run, counted on the communicator itself (device tensors over RCCL, host tensors over gloo)
        ones = torch.ones(1, dtype=torch.float64, device=dev if args.dist_backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones.item())))
        if ranks_seen != args.gpus:
            sys.stderr.write("bench.py: rank %d: the communicator holds %d ranks, --gpus %d\n" % (rank, ranks_seen, args.gpus))
            sys.exit(2)
    else:
        ranks_seen = 1

    out = run_config(args, rank, world, local_rank, headline_extras=True)

    # Short legs of the other BASELINE configs, after the headline's timed region and never part of its `value`: the driver only ever
    # runs the default command, and these make configs 3 / 4 / 5 measurements instead of claims.
    default_run = (args.config == 2 and not args.pairs and not args.size and not args.kerhw and args.dk < 0 and args.db < 0 and not args.no_other_configs)
    if default_run:
        legs = {}
        plan_of = {3: dict(config=3, pairs=0, steps=5, warmup=3), 5: dict(config=5, pairs=0, steps=5, warmup=3),
                   4: dict(config=2, pairs=62, steps=5, warmup=3)}
        order = [4] if world > 1 else [4, 3, 5]           # configs 3 and 5 are single-GPU configs; config 4 is the sharded batch
        for cid in order:
            a = copy.copy(args)
            a.batch, a.streams = 0, 0
            for k, v in plan_of[cid].items():
                setattr(a, k, v)
            try:
                leg = run_config(a, rank, world, local_rank, headline_extras=False)
            except Exception as e:      # a failing leg must not take the headline line with it
                leg = {"error": "%s: %s" % (type(e).__name__, e)} if rank == 0 else None
                if world > 1:
                    raise
            if rank == 0:
                legs[str(cid)] = leg
        if rank == 0:
            out["other_configs"] = dict(legs, note="short legs run after the headline's timed region (3 warm-up + 5 timed steps each, own "
                                        "barriers, plans and data); not part of `value`")
    if rank == 0:
        out["ranks_seen"] = ranks_seen
        out["launch"] = ("self" if os.environ.get("SFFT_BENCH_SELF_LAUNCHED") else "torchrun") + ":" + args.dist_backend if launched else "direct"
        emit(out)
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
