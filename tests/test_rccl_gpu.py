"""GPU: RCCL itself, at world_size 1 on the one-GPU box -- the library loads, the communicator is created on the device, and every
collective the bench and the sharded runner issue (`barrier`, `all_reduce` SUM / MAX, `all_gather` inside gather_records) runs once on
device tensors.  In a child process under a timeout: a communicator that hangs must fail this test, not the session.
Reference analogue: sfft/MultiEasyCrowdedPacket.py:361-399 (per-device workers); there the devices exchange nothing either."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from sfft_amd.sharding import pack_record, gather_records
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
dist.barrier()
one = torch.ones(1, dtype=torch.float64, device=dev)
dist.all_reduce(one, op=dist.ReduceOp.SUM)
t = torch.tensor([3.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
recs = [pack_record(i, 0, 1.5 * i, torch.arange(7, dtype=torch.float64, device=dev) + i) for i in range(3)]
table = gather_records(recs, 3, 7, dev)
parts = [torch.empty(5, dtype=torch.float64, device=dev)]
dist.all_gather(parts, torch.arange(5, dtype=torch.float64, device=dev))
torch.cuda.synchronize()
assert int(one.item()) == 1 and float(t.item()) == 3.25 and table.is_cuda and table.shape == (3, 10)
assert table[:, 0].tolist() == [0.0, 1.0, 2.0] and table[2, 3:].tolist() == [float(i + 2) for i in range(7)]
assert parts[0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version())
''' % ROOT


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.gpu
def test_rccl_world_size_one_runs_the_collectives_the_bench_uses():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", CHILD, str(_free_port())], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert "RCCL_OK" in r.stdout.decode()
