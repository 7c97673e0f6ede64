"""Multi-GPU use of the hot path: independent image pairs sharded over ranks, one process per GPU.

The reference's only multi-GPU mechanism is task parallelism over independent pairs (one Python thread per
device pulling tasks from a shared table, sfft/MultiEasyCrowdedPacket.py:361-399, 698-710); nothing is
exchanged between GPUs.  Here the pairs of a batch are dealt round-robin to the ranks of a torch.distributed
job and the only collective is an all_gather of one fixed-size record per pair
    [pair_id, status, milliseconds, Solution[NEQ]]
(RCCL over xGMI on GPUs, gloo on CPU in the tests).  Difference images never leave their GPU.
"""
import torch
import torch.distributed as dist

__all__ = ["shard_pair_ids", "pack_record", "gather_records"]


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
