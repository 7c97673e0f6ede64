"""GPU: bench.py's own N > 1 code path (process group, barriers, max-over-ranks timing, uneven shards, the gather of the records,
rank 0's single JSON line) launched the way the driver launches it -- `python -m torch.distributed.run --nproc-per-node 2 ... bench.py
--gpus 2` -- with both ranks on cuda:0 over gloo (a one-GPU box; RCCL refuses two ranks on one device, so RCCL itself stays
unexercised here).  Reference analogue: sfft/MultiEasyCrowdedPacket.py:361-399, 698-710."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "512", "--steps", "1", "--warmup", "1",
           "--no-cpu", "--no-other-configs", "--no-host-arrays", "--dist-backend", "gloo", "--all-ranks-on-device0"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.startswith("{")]
    assert len(json_lines) == 1, lines                   # rank 0 prints ONE line, rank 1 nothing
    assert lines[-1] == json_lines[0] and len(json_lines[0]) < 4096
    return json.loads(json_lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_sharded_batch():
    d = _launch(["--pairs", "6"])                         # config-4 mode: 6 pairs dealt to 2 ranks, worker threads pull
    assert d["n_gpus"] == 2 and d["gathered_pairs"] == 6 and d["failed_pairs"] == 0 and d["value"] > 0
    assert d["scaling"] == "strong" and d["config"]["pairs_per_step"] == 6 and d["post_check"]["bitwise_equal"] is True
    assert abs(d["value"] - 6 * d["steps"] / d["config"]["timed_region_s"]) <= 1e-3 * d["value"]


@pytest.mark.gpu
def test_bench_two_ranks_weak_scaling_headline_path():
    d = _launch(["--batch", "3", "--streams", "2"])      # the headline's mode: every rank holds its own batch
    assert d["n_gpus"] == 2 and d["gathered_pairs"] == 6 and d["failed_pairs"] == 0 and d["value"] > 0
    assert d["scaling"] == "weak" and d["config"]["pairs_per_step"] == 6 and d["post_check"]["bitwise_equal"] is True
    for k in ("roofline", "roofline_hbm", "roofline_greek", "roofline_solve"):
        assert set(("bound", "kernel", "achieved", "peak", "frac", "traffic", "avg_ms")) <= set(d[k]), k
