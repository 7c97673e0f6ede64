"""Multi-GPU use of the hot path: independent image pairs sharded over ranks, one process per GPU.

The reference's only multi-GPU mechanism is task parallelism over independent pairs (one Python thread per
device pulling tasks from a shared table, sfft/MultiEasyCrowdedPacket.py:361-399, 698-710); nothing is
exchanged between GPUs.  Here the pairs of a batch are dealt round-robin to the ranks of a torch.distributed
job and the only collective is an all_gather of one fixed-size record per pair
    [pair_id, status, milliseconds, Solution[NEQ]]
(RCCL over xGMI on GPUs, gloo on CPU in the tests).  Difference images never leave their GPU.
"""
import itertools
import sys
import threading
import time
import traceback

import torch
import torch.distributed as dist

__all__ = ["shard_pair_ids", "pack_record", "gather_records", "run_shard", "STATUS_OK", "STATUS_SINGULAR", "STATUS_ERROR", "all_failed_with_error"]

# per-pair status in the gathered record: the C ABI's return code of the pair's sfft_subtract (include/sfft_amd.h):
# 0 ok, -1 invalid argument, -2 unsupported size, -3 HIP error, -4 singular system, -5 out of memory
STATUS_OK, STATUS_ERROR, STATUS_SINGULAR = 0, -1, -4


def shard_pair_ids(n_pairs, rank, world_size):
    """Round-robin: pair i goes to rank i % world_size (62 pairs on 8 ranks -> 6 ranks x 8 + 2 ranks x 7)."""
    return list(range(rank, n_pairs, world_size))


def pack_record(pair_id, status, ms, solution):
    sol = solution.detach().to(torch.float64).reshape(-1)
    head = torch.tensor([float(pair_id), float(status), float(ms)], dtype=torch.float64, device=sol.device)
    return torch.cat([head, sol])


def gather_records(local_records, n_pairs, neq, device):
    """all_gather of per-pair records; every rank returns the [n_pairs, 3+NEQ] table ordered by pair id.
    Ranks may hold different numbers of pairs: shards are padded to the largest one with pair_id = -1."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    width = 3 + neq
    max_local = (n_pairs + world - 1) // world
    buf = torch.full((max_local, width), -1.0, dtype=torch.float64, device=device)
    for k, rec in enumerate(local_records):
        buf[k] = rec.to(device)
    if world > 1:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        allrec = torch.cat(parts, dim=0)
    else:
        allrec = buf
    table = torch.zeros((n_pairs, width), dtype=torch.float64, device=device)
    seen = torch.zeros(n_pairs, dtype=torch.bool, device=device)
    for row in allrec:
        pid = int(row[0].item())
        if pid >= 0:
            table[pid] = row
            seen[pid] = True
    if not bool(seen.all()):
        raise RuntimeError("gather_records: missing pairs %s" % (torch.where(~seen)[0].tolist(),))
    return table


def run_shard(pair_ids, n_workers, work_fn, neq, device, errors=None):
    """Process this rank's shard with `n_workers` host threads that pull the next pair id from a shared queue -- the
    reference's scheme (one thread per device queue taking tasks from a shared status table,
    sfft/MultiEasyCrowdedPacket.py:361-399, 698-710), with several queues on one GPU.

    work_fn(worker_index, pair_id) -> Solution tensor [neq]; it raises on failure like the operators do:
    numpy.linalg.LinAlgError for a singular system, _lib.SfftError (with the ABI's return code) for every other status of the
    C ABI, a plain Exception('MeLOn ERROR: ...') from the Python layer.  As in the reference (`except Exception` per task,
    sfft/MultiEasyCrowdedPacket.py:344, 646) a failing pair never stops the shard: its record carries a status (the ABI's code,
    STATUS_SINGULAR, or STATUS_ERROR for any other Exception -- pair id, message and traceback go to stderr, and (pair id, message)
    to `errors` when a list is passed) and a zero solution, and the worker's plan goes on to the next pair.  So every rank always reaches the collective in gather_records.
    Only a non-Exception BaseException (KeyboardInterrupt, SystemExit) stops the workers and is re-raised here.
    Returns one record per pair, in shard order."""
    import numpy as np
    from ._lib import SfftError
    pair_ids = list(pair_ids)
    records = [None] * len(pair_ids)
    nxt = itertools.count()
    lock = threading.Lock()
    fatal = []

    def worker(wi):
        while not fatal:
            with lock:
                k = next(nxt)
            if k >= len(pair_ids):
                return
            pid = pair_ids[k]
            t0 = time.perf_counter()
            try:
                sol = work_fn(wi, pid)
                status = STATUS_OK
            except np.linalg.LinAlgError:
                sol, status = torch.zeros(neq, dtype=torch.float64, device=device), STATUS_SINGULAR
            except SfftError as e:
                sol, status = torch.zeros(neq, dtype=torch.float64, device=device), e.code
            except Exception as e:          # any other per-pair failure: recorded, the shard carries on -- never silently
                sol, status = torch.zeros(neq, dtype=torch.float64, device=device), STATUS_ERROR
                msg = "%s: %s" % (type(e).__name__, e)
                if errors is not None:
                    errors.append((pid, msg))
                # the reference prints the task's exception too (sfft/MultiEasyCrowdedPacket.py:344-351); with the traceback a
                # programming error or a sticky device fault is not mistaken for a bad pair
                sys.stderr.write("sfft_amd.sharding: pair %d failed on worker %d: %s\n%s" % (pid, wi, msg, traceback.format_exc()))
                sys.stderr.flush()
            except BaseException as e:      # KeyboardInterrupt / SystemExit: stop taking pairs
                fatal.append(e)
                return
            records[k] = pack_record(pid, status, (time.perf_counter() - t0) * 1e3, sol)

    if n_workers <= 1:
        worker(0)
    else:
        th = [threading.Thread(target=worker, args=(i,)) for i in range(n_workers)]
        [t.start() for t in th]
        [t.join() for t in th]
    if fatal:
        raise fatal[0]
    return records


def all_failed_with_error(records):
    """True when a non-empty shard's pairs ALL ended with STATUS_ERROR (an Exception that is neither a singular system nor an ABI
    status): that is a broken worker, not a batch of bad pairs.  Callers raise AFTER gather_records so that every rank still
    reaches the collective."""
    return len(records) > 0 and all(int(r[1].item()) == STATUS_ERROR for r in records)
