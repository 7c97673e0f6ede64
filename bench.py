#!/usr/bin/env python3
"""bench.py -- headline benchmark of the SFFT subtraction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one GSS-equivalent pass (solve on the masked pair + apply to the full pair, the body of
sfft.PureCupy_Customized_Packet.PCCP) over one synthetic 4096 x 4096 image pair, KerHW 8, KerPolyOrder 2,
BGPolyOrder 2, ConstPhotRatio, fp64 -- BASELINE.json configs[1].  Inputs are resident in HBM when the timed
region starts; the plan (tables + workspaces) is created before it.  With N ranks every rank runs its own pair
per step (weak scaling, independent pairs, no data-path collective); the only collective is the gather of
per-pair records at the end (sfft_amd/sharding.py).

Rank 0 prints ONE JSON line.  `value` = image pairs per second over all ranks.  Extra objects:
  roofline     -- the dominant stage (by HIP-event time measured inside the timed region, on the stream the
                  kernels run on): algorithmic bytes / average duration against the 8 TB/s HBM3E peak
  cpu_baseline -- the numpy/scipy oracle (port of the reference's Numpy backend) timed on this host on a
                  bounded sample (smaller image, same kernel geometry), converted to 4096^2-pairs/s by pixel count
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


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
    out = {
        "prelim_solve": (Fij + 1) * fwd_plane + r * P,  # + row moments of J
        "greek_g1": (n_omg + n_the) * 2 * spec + n_gam * spec,   # operands A and B (B generated on the fly for Gamma)
        "prelim_apply": Fij * fwd_plane,
        "construct": Fij * spec + spec,
        "inverse": 2 * spec + spec + r * P + r * P,      # columns r+w, rows read, J read, DIFF write
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
    ap.add_argument("--cpu-sample", type=int, default=2048, help="side of the CPU-baseline sample image (0 = skip)")
    args = ap.parse_args()

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

    N = args.size
    t0 = time.perf_counter()
    plan = get_plan(N, N, args.kerhw, args.dk, args.db, True, local_rank)
    torch.cuda.synchronize(dev)
    plan_s = time.perf_counter() - t0
    pair = make_pair(N, N, seed=1234 + rank, mask=True, sky=0.0, bkg_scale=0.05)
    g = {k: torch.from_numpy(v).to(dev) for k, v in pair.items()}
    sol = torch.empty(plan.NEQ, dtype=torch.float64, device=dev)
    diff = torch.empty((N, N), dtype=torch.float64, device=dev)

    def step():
        plan.subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"], out_solution=sol, out_diff=diff)

    for _ in range(args.warmup):
        step()
    plan.set_timing(True)
    stage_acc = {}

    def barrier():
        if world > 1:
            dist.barrier()
    torch.cuda.synchronize(dev)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in plan.stage_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t_start
    plan.set_timing(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # gather one record per pair (pair id = rank): the only collective of the data path
    rec = pack_record(rank, 0, elapsed * 1e3 / max(args.steps, 1), sol)
    table = gather_records([rec], world, plan.NEQ, dev)

    if rank == 0:
        pairs = world * args.steps
        value = pairs / elapsed
        ms_step = elapsed * 1e3 / args.steps
        stage_ms = {k: v / args.steps for k, v in stage_acc.items()}
        ab = alg_bytes(N, N, args.kerhw, args.dk, args.db)
        timed = {k: v for k, v in stage_ms.items() if k in ab}
        dom = max(timed, key=timed.get)
        ach = ab[dom] / (timed[dom] * 1e-3) / 1e9
        out = {
            "metric": "image-pairs/sec, %dx%d, KerHW=%d polyOrd=%d" % (N, N, args.kerhw, args.dk),
            "value": value, "unit": "image-pairs/s", "mpix_per_s": value * N * N / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: one %dx%d pair per step per GPU, KerHW %d, KerPolyOrder %d, "
                                   "BGPolyOrder %d, ConstPhotRatio, fp64; GSS = solve(masked pair) + apply(full pair)"
                                   % (N, N, args.kerhw, args.dk, args.db),
                       "pairs_per_step": world, "plan_create_s": plan_s, "solver": {1: "cholesky", 2: "lu"}.get(plan.query("LAST_SOLVER"), "?")},
            "stage_ms": stage_ms,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "alg_bytes_per_launch": ab[dom], "avg_ms": timed[dom]},
            "pair_effective": {"B_alg_reference_bytes": ab["B_alg_reference"], "n_fft_reference": ab["n_fft_reference"],
                               "effective_GBs": ab["B_alg_reference"] / (ms_step * 1e-3 / world) / 1e9 / world,
                               "note": "reference-algorithm bytes (SURVEY 8d) / measured pair time; the build moves fewer bytes"},
            "gathered_pairs": int(table.shape[0]),
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.kerhw, args.dk, args.db, N * N, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
