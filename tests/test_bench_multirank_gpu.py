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


SMALL = ["--size", "512", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-other-configs", "--no-host-arrays"]


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _launch(extra, torchrun=True, gpus=2, tail=("--dist-backend", "gloo", "--all-ranks-on-device0"), env_extra=None):
    """torchrun=True: the way the driver launches N > 1; False: plain `python bench.py --gpus N`, which must start its ranks itself."""
    head = [sys.executable]
    if torchrun:
        head += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd = head + [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + SMALL + list(tail) + extra
    r = subprocess.run(cmd, cwd=ROOT, env=dict(_clean_env(), **(env_extra or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
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
    _check_per_rank(d, [3, 3])
    assert abs(d["value"] - 6 * d["steps"] / d["config"]["timed_region_s"]) <= 1e-3 * d["value"]


def _check_per_rank(d, pairs_of_rank):
    """The line carries one row per rank -- rank, device, pairs of the timed region, that rank's own pairs/s, the NUMA node of its GPU and
    the CPUs its threads may run on -- and the timed region's solver accounting (sfft/MultiEasyCrowdedPacket.py:361-399, 646-669 prints
    per-task outcomes per device; one aggregate number would hide an imbalanced or badly placed rank)."""
    assert d["per_rank_keys"] == "rank,device,pairs,pairs_per_s,numa_node,cpus_allowed"
    rows = d["per_rank"]
    assert [r[0] for r in rows] == list(range(len(pairs_of_rank))) and [r[2] for r in rows] == [p * d["steps"] for p in pairs_of_rank]
    assert all(r[3] > 0 and r[5] >= 1 for r in rows)
    # every rank's own rate is at least the job's share of it (the job's clock is the slowest rank's, barrier included)
    assert all(r[3] >= 0.999 * d["value"] * r[2] / (sum(pairs_of_rank) * d["steps"]) for r in rows)
    assert d["solves_timed"] == sum(pairs_of_rank) * d["steps"]


@pytest.mark.gpu
def test_bench_two_ranks_weak_scaling_headline_path():
    d = _launch(["--batch", "3", "--streams", "2"])      # the headline's mode: every rank holds its own batch
    assert d["n_gpus"] == 2 and d["gathered_pairs"] == 6 and d["failed_pairs"] == 0 and d["value"] > 0
    assert d["scaling"] == "weak" and d["config"]["pairs_per_step"] == 6 and d["post_check"]["bitwise_equal"] is True
    _check_per_rank(d, [3, 3])
    assert d["lu_fallback_pairs"] == 0 and d["chol_stall_events"] == 0 and d["config"]["solver"] == "cholesky"
    for k in ("roofline", "roofline_hbm", "roofline_greek", "roofline_solve"):
        assert set(("bound", "kernel", "achieved", "peak", "frac", "traffic", "avg_ms")) <= set(d[k]), k


@pytest.mark.gpu
def test_bench_line_counts_lu_fallbacks_of_the_timed_region():
    """SFFT_TEST_FAIL_CHOL=1 reports every Cholesky attempt as failed, so every pair of the timed region is redone by the pivoted LU: the line
    must say so (`lu_fallback_pairs` == the solves of the timed region, `config.solver` names the fallback) instead of "cholesky"."""
    d = _launch(["--pairs", "5"], env_extra={"SFFT_TEST_FAIL_CHOL": "1"})
    _check_per_rank(d, [3, 2])
    assert d["lu_fallback_pairs"] == d["solves_timed"] == 5 * d["steps"] and d["failed_pairs"] == 0
    assert "LU fallback on %d of %d" % (d["lu_fallback_pairs"], d["solves_timed"]) in d["config"]["solver"]


@pytest.mark.gpu
def test_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it and no WORLD_SIZE in the environment: two ranks run, the line says so."""
    d = _launch(["--pairs", "5"], torchrun=False)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["launch"] == "self:gloo"
    assert d["gathered_pairs"] == 5 and d["failed_pairs"] == 0 and d["config"]["pairs_per_step"] == 5


@pytest.mark.gpu
def test_one_rank_through_the_spawner_over_rccl():
    """--gpus 1 --spawn: one rank started through torch.distributed.run with the nccl (= RCCL) backend: communicator created on the device,
    barrier / all_reduce(MAX) / the record gather run on device tensors."""
    d = _launch(["--batch", "3", "--streams", "2", "--spawn"], torchrun=False, gpus=1, tail=("--dist-backend", "nccl"))
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["launch"] == "self:nccl"
    assert d["gathered_pairs"] == 3 and d["failed_pairs"] == 0 and d["post_check"]["bitwise_equal"] is True


def _expect_refusal(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr.decode()[-2000:])
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]        # no JSON line: never a smaller job's number
    return r.stderr.decode()


def test_world_size_mismatch_fails_instead_of_warning():
    """(no GPU needed: the check runs before any device work)  WORLD_SIZE=1 in the environment with --gpus 2 is an error, not a 1-GPU line."""
    env = dict(_clean_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    err = _expect_refusal([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, env)
    assert "WORLD_SIZE=1 but --gpus 2" in err


def test_more_gpus_than_the_node_has_is_refused():
    import torch
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    err = _expect_refusal([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 2)] + SMALL, _clean_env())
    assert "refusing" in err
