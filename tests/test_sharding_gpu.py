"""GPU: the sharded batch runner (config 4, bench.py --pairs) across PROCESSES on the HIP path -- world_size 2 over gloo, both
ranks on cuda:0, real plans, worker threads pulling from the shard's queue, one singular pair, records gathered by all_gather.
The gathered table must equal a single-process run of the same batch (every kernel is deterministic).  This is the reference's
multi-task scheme (sfft/MultiEasyCrowdedPacket.py:361-399, 698-710: one queue per device, failed tasks recorded and skipped)
with processes instead of threads; an 8-GPU node is not needed to exercise it."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N0, N1, W, DK, DB = 192, 160, 3, 2, 1
N_PAIRS, BAD = 7, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_batch(rank, world):
    """This rank's shard of the batch on cuda:0 with two workers (plan + stream each); returns the gathered table (CPU)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sfft_amd.plan import Plan
    from sfft_amd.sharding import shard_pair_ids, gather_records, run_shard
    from sfft_amd.utils.synthetic import make_pair
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    plans = [Plan(N0, N1, W, DK, DB, True, device=0) for _ in range(2)]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    NEQ = plans[0].NEQ
    my_ids = shard_pair_ids(N_PAIRS, rank, world)
    data = {}
    for pid in my_ids:
        if pid == BAD:
            z = torch.zeros((N0, N1), dtype=torch.float64, device=dev)
            data[pid] = dict(REF=z, SCI=z.clone(), mREF=z.clone(), mSCI=z.clone())        # no signal at all: a singular system
        else:
            pr = make_pair(N0, N1, seed=200 + pid, mask=True, density=400.0)
            data[pid] = {k: torch.from_numpy(v).to(dev) for k, v in pr.items()}

    def work(wi, pid):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[wi]):
            g = data[pid]
            sol, _ = plans[wi].subtract(g["REF"], g["SCI"], g["mREF"], g["mSCI"])
            streams[wi].synchronize()
            return sol
    recs = run_shard(my_ids, 2, work, NEQ, dev)
    torch.cuda.synchronize(dev)
    return gather_records([r.cpu() for r in recs], N_PAIRS, NEQ, torch.device("cpu")).numpy()


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        np.save(os.path.join(out_dir, "table_%d.npy" % rank), _run_batch(rank, world))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_processes_on_one_gpu_match_a_single_process_run(tmp_path):
    from sfft_amd.sharding import STATUS_OK, STATUS_SINGULAR
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    t0, t1 = np.load(tmp_path / "table_0.npy"), np.load(tmp_path / "table_1.npy")
    assert np.array_equal(t0, t1)                           # every rank ends with the same table
    single = _run_batch(0, 1)                               # the whole batch in this process, no process group
    assert t0.shape == single.shape and list(t0[:, 0]) == list(range(N_PAIRS))
    assert np.array_equal(t0[:, 1], single[:, 1])
    for pid in range(N_PAIRS):
        if pid == BAD:
            assert t0[pid, 1] == STATUS_SINGULAR and not t0[pid, 3:].any()
        else:
            assert t0[pid, 1] == STATUS_OK and np.isfinite(t0[pid, 3:]).all() and t0[pid, 3:].any()
        assert t0[pid, 2] > 0.0
    assert np.array_equal(t0[:, 3:], single[:, 3:])         # solutions bit for bit
