#!/usr/bin/env python3
"""bench.py -- headline benchmark of the SFFT subtraction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one GSS-equivalent pass (solve on the masked pair + apply to the full pair, the body of
sfft.PureCupy_Customized_Packet.PCCP) over one batch of synthetic 4096 x 4096 image pairs, KerHW 8,
KerPolyOrder 2, BGPolyOrder 2, ConstPhotRatio, fp64 -- BASELINE.json configs[1].  The batch is `--streams` pairs
per GPU (default 3), each on its own plan, HIP stream and host thread: the dense solve of one pair is latency
bound and leaves most CUs idle, so independent pairs are pipelined on one GPU exactly as the reference's
multi-task packet pipelines them with one thread per device queue.  Inputs are resident in HBM when the timed
region starts; plans (tables + workspaces) are created before it.  With N ranks every rank runs its own batch
per step (weak scaling, no data-path collective); the only collective is the gather of per-pair records at the
end (sfft_amd/sharding.py).

Rank 0 prints ONE JSON line.  `value` = image pairs per second over all ranks.  Extra objects:
  roofline     -- the dominant KERNEL by time per pair.  Since the forward transforms were halved that is the Omega pass of the
                  Greek stage (greek_g1_mfma<2,false>: v_mfma_f64_16x16x4_f64), priced against the fp64 MFMA peak (bound "mfma"); `roofline_hbm` is the dominant HBM-bound kernel, the forward column pass
                  (cols_fwd_weighted_4096_q): algorithmic bytes of the timed launch / its duration (HIP events on the launch
                  stream) against the 8 TB/s HBM3E peak; `roofline_greek` is the same for the second kernel, the Omega
                  pass of the Greek stage, which is bound by fp64 FMA issue, not by HBM
  cpu_baseline -- the numpy/scipy oracle (port of the reference's Numpy backend) timed on this host on a
                  bounded sample (smaller image, same kernel geometry), converted to 4096^2-pairs/s by pixel count
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

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6    # same guide: fp64 vector = fp64 matrix (MFMA) peak on MI355X


def alg_bytes(N0, N1, w, DK, DB):
    """Algorithmic HBM bytes per stage for ONE pair (solve + apply), as built, and the
    canonical reference-algorithm figure B_alg of SURVEY.md 8(d)."""
    P = N0 * N1
    Nh = N1 // 2 + 1
    Fij = (DK + 1) * (DK + 2) // 2
    Fpq = (DB + 1) * (DB + 2) // 2
    c, r = 16, 8
    spec = c * N0 * Nh                                  # one half-spectrum plane
    fwd_plane = r * P + spec + 2 * spec                 # rows: read image, write spectrum; columns: read + write
    n_omg, n_gam, n_the = Fij * (Fij + 1) // 2, Fij * Fpq, Fij
    n_gamp = Fij * DB                                   # dense Gamma column-factor passes (p >= 1); they read A only
    out = {
        "prelim_solve": (Fij + 1) * fwd_plane + r * P,  # + row moments of J
        # forward transforms of the solve pass as built: one row transform per distinct column factor (DK + 1 of them, + J) into
        # stage planes; the column pass reads every stage plane (from HBM once, its other readers hit L2) and writes Fij + 1 planes
        "fwd_rows": 2 * r * P + (DK + 2) * spec,
        "fwd_cols": (DK + 2) * spec + (Fij + 1) * spec,
        # Greek stage 1 as built: all passes of a (64-column x row-chunk) tile run on one XCD back to back, so each of the
        # Fij (+ J) planes is streamed from HBM once and re-read from that XCD's L2; partial lag sums are written
        "greek_g1": Fij * spec,
        "greek_g1b": (Fij + 1) * spec + 8 * N0 * N1,      # Theta passes: Fij planes + FJ; Gamma block: one read of the masked image
        # fp64 flops of the Omega passes: per pass and spectrum element one complex product (6) + 4 real FMAs per lag
        # (the Fij diagonal passes have a real product: half the lag work)
        "greek_g1_flops": N0 * Nh * ((n_omg - Fij) * (6 + 2 * 4 * (2 * w)) + Fij * (3 + 4 * (2 * w))),
        # apply pass as built (polynomial kernel, KerHW <= 8): row pass into DK + 1 stage planes, mixed-domain column convolution
        # (reads them, writes one plane), inverse row pass with the DIFF epilogue -- no column transforms
        "prelim_apply": r * P + (DK + 1) * spec,
        "construct": (DK + 1) * spec + spec,
        "inverse": spec + r * P + r * P,                # rows read, J read, DIFF write
    }
    n_pre = 1 + Fij + Fpq
    n_greek = Fij * Fij + 2 * Fij * Fpq + Fpq * Fpq + Fij + Fpq
    n_fft = 2 * n_pre + n_greek + 1
    out["B_alg_reference"] = n_fft * 4 * c * P + n_greek * c * P + n_pre * c * P + 5 * r * P
    out["n_fft_reference"] = n_fft
    return out


def cpu_baseline(w, DK, DB, full_pixels, sample_side):
    """Time the oracle's GSS on a bounded sample: a sample_side^2 pair with the same kernel geometry."""
    from oracle import sfft_oracle as O
    from sfft_amd.utils.synthetic import make_pair
    cores = os.cpu_count() or 1
    pair = make_pair(sample_side, sample_side, seed=4321, mask=True, sky=0.0, bkg_scale=0.05)
    p = O.SSC(sample_side, sample_side, w, DK, DB, True)
    t0 = time.perf_counter()
    O.GSS(pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], p, workers=cores)
    dt = time.perf_counter() - t0
    mpix_s = sample_side * sample_side / 1e6 / dt
    return {"value": mpix_s * 1e6 / full_pixels, "unit": "image-pairs/s", "mpix_per_s": mpix_s, "cores": cores,
            "kind": "port",
            "sample": "one GSS (solve+apply) on a %dx%d synthetic pair, KerHW %d, orders %d/%d, numpy oracle with "
                      "scipy.fft workers=%d; %.1f s; scaled to 4096^2 pairs by pixel count"
                      % (sample_side, sample_side, w, DK, DB, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--kerhw", type=int, default=8)
    ap.add_argument("--dk", type=int, default=2)
    ap.add_argument("--db", type=int, default=2)
    ap.add_argument("--streams", type=int, default=4,
                    help="independent pairs in flight per GPU (one plan + stream + host thread each); a step = this many pairs")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="side of the CPU-baseline sample image (0 = skip)")
    args = ap.parse_args()

    import threading
    import torch
    import torch.distributed as dist
    from sfft_amd.plan import get_plan
    from sfft_amd.sharding import pack_record, gather_records
    from sfft_amd.utils.synthetic import make_pair

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("warning: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    N, S = args.size, max(1, args.streams)
    t0 = time.perf_counter()
    plans = [get_plan(N, N, args.kerhw, args.dk, args.db, True, local_rank, slot=i) for i in range(S)]
    torch.cuda.synchronize(dev)
    plan_s = (time.perf_counter() - t0) / S
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    # one synthetic pair per stream (pair id = rank * S + i), resident in HBM before the timed region
    pairs = [make_pair(N, N, seed=1234 + rank * S + i, mask=True, sky=0.0, bkg_scale=0.05) for i in range(S)]
    g = [{k: torch.from_numpy(v).to(dev) for k, v in pr.items()} for pr in pairs]
    sols = [torch.empty(plans[0].NEQ, dtype=torch.float64, device=dev) for _ in range(S)]
    diffs = [torch.empty((N, N), dtype=torch.float64, device=dev) for _ in range(S)]
    stage_acc = {}

    def run(i, n, timed):
        torch.cuda.set_device(local_rank)
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                plans[i].subtract(g[i]["REF"], g[i]["SCI"], g[i]["mREF"], g[i]["mSCI"], out_solution=sols[i], out_diff=diffs[i])
                if timed and i == 0:
                    for k, v in plans[0].stage_ms().items():
                        stage_acc[k] = stage_acc.get(k, 0.0) + v

    def run_all(n, timed):
        if S == 1:
            run(0, n, timed)
            return
        th = [threading.Thread(target=run, args=(i, n, timed)) for i in range(S)]
        [t.start() for t in th]
        [t.join() for t in th]

    run_all(args.warmup, False)
    plans[0].set_timing(True)

    def barrier():
        if world > 1:
            dist.barrier()
    torch.cuda.synchronize(dev)
    barrier()
    t_start = time.perf_counter()
    run_all(args.steps, True)
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t_start

    # isolated pass (one pair in flight) for per-kernel roofline numbers that are not inflated by the other streams
    iso_acc = {}
    n_iso = min(5, max(2, args.steps))
    with torch.cuda.stream(streams[0]):
        t_iso = time.perf_counter()
        for _ in range(n_iso):
            plans[0].subtract(g[0]["REF"], g[0]["SCI"], g[0]["mREF"], g[0]["mSCI"], out_solution=sols[0], out_diff=diffs[0])
            for k, v in plans[0].stage_ms().items():
                iso_acc[k] = iso_acc.get(k, 0.0) + v
        torch.cuda.synchronize(dev)
        iso_ms = (time.perf_counter() - t_iso) * 1e3 / n_iso
    plans[0].set_timing(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # gather one record per pair: the only collective of the data path
    recs = [pack_record(rank * S + i, 0, elapsed * 1e3 / max(args.steps, 1), sols[i]) for i in range(S)]
    table = gather_records(recs, world * S, plans[0].NEQ, dev)

    if rank == 0:
        npairs = world * S * args.steps
        value = npairs / elapsed
        ms_step = elapsed * 1e3 / args.steps
        stage_ms = {k: v / args.steps for k, v in stage_acc.items()}
        iso_stage = {k: v / n_iso for k, v in iso_acc.items()}
        ab = alg_bytes(N, N, args.kerhw, args.dk, args.db)

        pmc = {}
        try:   # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/pmc_traffic.json; see profiles/README.md)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            pass

        KERNEL_OF = {"fwd_cols": "cols_fwd_weighted_4096_q" if N == 4096 else "cols_fwd_weighted / strided_dft",
                     "fwd_rows": "rows_r2c_4096" if N == 4096 else "rows_r2c", "greek_g1": "greek_g1_mfma<2, false> (Omega passes)",
                     "greek_g1b": "greek_g1<8, 2> (Theta passes) + row_moments / gamma_rows / gamma_patches (Gamma block)", "construct": "construct_fd"}

        def roof(stages, dom="fwd_cols"):
            ach = ab[dom] / (stages[dom] * 1e-3) / 1e9
            traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch") if (N, args.kerhw, args.dk, args.db) == (4096, 8, 2, 2) else None
            return {"bound": "hbm", "kernel": KERNEL_OF.get(dom, dom), "stage": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "sustained_peak_measured": 5000.0,   # profiles/r01_hbm_stream.txt: a plain copy kernel on this device, GB/s
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "alg_bytes_per_launch": ab[dom], "avg_ms": stages[dom]}

        def roof_flops(stages):
            tf = ab["greek_g1_flops"] / (stages["greek_g1"] * 1e-3) / 1e12
            traffic = pmc.get("greek_g1", {}).get("hbm_bytes_per_launch") if (N, args.kerhw, args.dk, args.db) == (4096, 8, 2, 2) else None
            return {"bound": "mfma", "kernel": KERNEL_OF["greek_g1"], "stage": "greek_g1", "achieved": tf, "peak": FP64_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS, "traffic": traffic, "alg_flops_per_launch": ab["greek_g1_flops"],
                    "alg_bytes_per_launch": ab["greek_g1"], "avg_ms": stages["greek_g1"],
                    "sustained_peak_measured": 47.4,     # profiles/r01_mfma_f64_peak.txt: a loop of independent MFMAs, TFLOP/s
                    "note": "v_mfma_f64_16x16x4_f64 (the fp64 matrix and vector peaks are equal on MI355X); a loop of nothing but independent "
                            "MFMAs sustains 47.4 TFLOP/s on this device (scripts/micro/mfma_f64_peak.hip); HBM side: "
                            "%.0f GB/s of algorithmic bytes" % (ab["greek_g1"] / (stages["greek_g1"] * 1e-3) / 1e9)}
        out = {
            "metric": "image-pairs/sec, %dx%d, KerHW=%d polyOrd=%d" % (N, N, args.kerhw, args.dk),
            "value": value, "unit": "image-pairs/s", "mpix_per_s": value * N * N / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %dx%d pairs, KerHW %d, KerPolyOrder %d, BGPolyOrder %d, "
                                   "ConstPhotRatio, fp64; GSS = solve(masked pair) + apply(full pair); "
                                   "%d independent pairs in flight per GPU (one plan + stream each), a step = %d pairs"
                                   % (N, N, args.kerhw, args.dk, args.db, S, world * S),
                       "pairs_per_step": world * S, "pairs_in_flight_per_gpu": S, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "plan_create_s": plan_s,
                       "solver": {1: "cholesky", 2: "lu"}.get(plans[0].query("LAST_SOLVER"), "?")},
            "stage_ms": stage_ms,
            # the dominant kernel by time per pair: the Omega pass of the Greek stage (one launch per pair) since the forward
            # transforms were halved; it is bound by fp64 FMA issue.  The dominant HBM-bound kernel (forward column pass) follows.
            "roofline": dict(roof_flops(iso_stage) if iso_stage["greek_g1"] >= iso_stage["fwd_cols"] else roof(iso_stage),
                             measured="HIP events on the launch stream around the kernel, %d launches with one pair in flight right after "
                             "the timed region (same process, same buffers)" % n_iso,
                             kernel_ms_per_pair={k: iso_stage[k] for k in ("fwd_rows", "fwd_cols", "greek_g1", "greek_g1b", "construct")}),
            "roofline_hbm": dict(roof(iso_stage), measured="same events, same launches (the forward column launch of the solve pass: "
                                 "%d stage planes in, %d planes out)" % (args.dk + 2, (args.dk + 1) * (args.dk + 2) // 2 + 1)),
            "roofline_greek": dict(roof_flops(iso_stage), measured="same events, same launches"),
            "roofline_timed_region": dict(roof(stage_ms), measured="same events on stream 0 inside the timed region; durations "
                                          "include time sliced to the other %d streams' kernels" % (S - 1)),
            "single_pair": {"ms": iso_ms, "pairs_per_s": 1e3 / iso_ms, "stage_ms": iso_stage,
                            "note": "one pair in flight: latency of one GSS and per-stage times without interleaving"},
            "pair_effective": {"B_alg_reference_bytes": ab["B_alg_reference"], "n_fft_reference": ab["n_fft_reference"],
                               "effective_GBs_per_gpu": ab["B_alg_reference"] * (value / world) / 1e9,
                               "as_built_bytes_per_pair": sum(ab[k] for k in ("fwd_rows", "fwd_cols", "greek_g1", "greek_g1b", "prelim_apply", "construct", "inverse")),
                               "note": "reference-algorithm bytes (SURVEY 8d) x pairs/s per GPU; context, not the roofline: the build's own "
                                       "algorithmic bytes per pair are listed beside it"},
            "gathered_pairs": int(table.shape[0]),
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.kerhw, args.dk, args.db, N * N, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
