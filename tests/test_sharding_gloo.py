"""CPU: the N>1 path (pair sharding + record gather) with world_size 2 over gloo.  The compute inside each
rank is the oracle on tiny pairs (test-only stand-in for the HIP call; the sharding layer is what is tested)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sfft_amd.sharding import shard_pair_ids, pack_record, gather_records, run_shard, STATUS_OK, STATUS_SINGULAR


def test_round_robin_sharding_covers_all_pairs():
    for n, w in [(62, 8), (5, 2), (3, 4), (16, 1)]:
        ids = [shard_pair_ids(n, r, w) for r in range(w)]
        flat = sorted(i for s in ids for i in s)
        assert flat == list(range(n))
        assert max(len(s) for s in ids) - min(len(s) for s in ids) <= 1
    assert [len(shard_pair_ids(62, r, 8)) for r in range(8)] == [8, 8, 8, 8, 8, 8, 7, 7]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _solve_pair(pid):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import sfft_oracle as O
    from sfft_amd.utils.synthetic import make_pair
    pair = make_pair(32, 24, seed=100 + pid, mask=False)
    p = O.SSC(32, 24, 1, 0, 0, True)
    sol, _ = O.ESS(pair["mREF"], pair["mSCI"], p)
    return sol


def _worker(rank, world, port, n_pairs, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        recs = []
        for pid in shard_pair_ids(n_pairs, rank, world):
            sol = torch.from_numpy(_solve_pair(pid))
            recs.append(pack_record(pid, 0, 1.5 + pid, sol))
        table = gather_records(recs, n_pairs, 10, torch.device("cpu"))
        np.save(os.path.join(out_dir, "table_%d.npy" % rank), table.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_matches_single_process(tmp_path):
    n_pairs, world = 5, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_pairs, str(tmp_path)), nprocs=world, join=True)
    t0 = np.load(tmp_path / "table_0.npy")
    t1 = np.load(tmp_path / "table_1.npy")
    assert np.array_equal(t0, t1)                       # every rank ends with the same table
    assert t0.shape == (n_pairs, 13)
    for pid in range(n_pairs):
        assert t0[pid, 0] == pid and t0[pid, 1] == 0 and t0[pid, 2] == 1.5 + pid
        assert np.array_equal(t0[pid, 3:], _solve_pair(pid))


def test_single_process_gather_without_process_group():
    recs = [pack_record(i, 0, 2.0, torch.arange(4, dtype=torch.float64) + i) for i in range(3)]
    table = gather_records(recs, 3, 4, torch.device("cpu"))
    assert table.shape == (3, 7) and table[2, 3] == 2.0
    with pytest.raises(RuntimeError, match="missing pairs"):
        gather_records(recs[:2], 3, 4, torch.device("cpu"))


# ---------------------------------------------------------------------------------------------------
# config-4 batch mode (bench.py --pairs M): uneven shards, worker threads pulling from the shard's queue, real status codes
# ---------------------------------------------------------------------------------------------------
def _batch_worker(rank, world, port, n_pairs, bad_pair, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = []

        def work(wi, pid):
            seen.append((wi, pid))
            if pid == bad_pair:
                raise np.linalg.LinAlgError("Singular matrix")
            return torch.from_numpy(_solve_pair(pid))
        recs = run_shard(shard_pair_ids(n_pairs, rank, world), 2, work, 10, torch.device("cpu"))
        assert sorted(p for _, p in seen) == shard_pair_ids(n_pairs, rank, world)
        table = gather_records(recs, n_pairs, 10, torch.device("cpu"))
        np.save(os.path.join(out_dir, "batch_%d.npy" % rank), table.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_batch_with_uneven_shards_and_a_failed_pair(tmp_path):
    n_pairs, world, bad = 7, 2, 4            # shards of 4 and 3 pairs; pair 4 fails and must not stop its shard
    port = _free_port()
    mp.spawn(_batch_worker, args=(world, port, n_pairs, bad, str(tmp_path)), nprocs=world, join=True)
    t0 = np.load(tmp_path / "batch_0.npy")
    assert np.array_equal(t0, np.load(tmp_path / "batch_1.npy"))
    assert t0.shape == (n_pairs, 13) and list(t0[:, 0]) == list(range(n_pairs))
    for pid in range(n_pairs):
        if pid == bad:
            assert t0[pid, 1] == STATUS_SINGULAR and not t0[pid, 3:].any()
        else:
            assert t0[pid, 1] == STATUS_OK and np.array_equal(t0[pid, 3:], _solve_pair(pid))
        assert t0[pid, 2] > 0.0              # measured milliseconds of that pair


def test_run_shard_records_every_exception_and_carries_on():
    """Per-pair failures never stop the shard (the reference's `except Exception` per task, MultiEasyCrowdedPacket.py:344, 646):
    LinAlgError -> -4, _lib.SfftError -> its code, any other Exception -> STATUS_ERROR with the message kept; only a
    non-Exception BaseException is re-raised."""
    from sfft_amd._lib import SfftError
    from sfft_amd.sharding import STATUS_ERROR

    def work(wi, pid):
        if pid == 1:
            raise SfftError(-3, "HIP error: out of memory")
        if pid == 2:
            raise np.linalg.LinAlgError("Singular matrix")
        if pid == 4:
            raise Exception("MeLOn ERROR: Input images should have the same shape!")
        if pid == 5:
            raise TypeError("a torch error on one pair")
        return torch.full((4,), float(pid), dtype=torch.float64)
    errors = []
    recs = run_shard([0, 1, 2, 3, 4, 5], 2, work, 4, torch.device("cpu"), errors=errors)
    assert [int(r[1]) for r in recs] == [0, -3, STATUS_SINGULAR, 0, STATUS_ERROR, STATUS_ERROR]
    assert not recs[1][3:].any() and recs[3][3] == 3.0 and not recs[4][3:].any()
    assert sorted(p for p, _ in errors) == [4, 5] and any("MeLOn ERROR" in m for _, m in errors)

    def interrupted(wi, pid):
        if pid == 2:
            raise KeyboardInterrupt()
        return torch.zeros(4, dtype=torch.float64)
    with pytest.raises(KeyboardInterrupt):
        run_shard([0, 1, 2, 3, 4, 5], 2, interrupted, 4, torch.device("cpu"))
