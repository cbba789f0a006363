"""The product's own data-parallel training step under a process group: two ranks (gloo, one MI355X shared by both; RCCL
needs a card per rank and the scaling run belongs to the driver) run `train_artifact_model` on their shards of a dataset.
Every optimizer step all-reduces the gradient in two buckets, the early one on a side stream under the rest of the backward
(training/distributed.py: BucketedGradAllReduce); replicas are made identical at the start by a broadcast and must STILL be
identical after two training epochs and a calibration epoch (whose torch optimizer steps only the calibration parameters)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    """a port nobody listens on right now (fixed ports collide under parallel pytest runs or with a stale worker)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_training_keeps_replicas_identical():
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dp_worker.py"), d]
        res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        r0, r1 = (torch.load(os.path.join(d, f"rank{r}.pt"), weights_only=False) for r in range(2))
    assert r0["hook"] == "BucketedGradAllReduce" and 0 < r0["late_start"] < r0["theta"].numel()
    assert torch.isfinite(r0["theta"]).all()
    assert torch.equal(r0["theta"], r1["theta"])          # bit-identical replicas after training AND calibration epochs
    assert r0["history"] == r1["history"]                 # the all-reduced epoch statistics agree too
    assert [h[:2] for h in r0["history"]] == [(1, "TRAIN"), (1, "VALID"), (2, "TRAIN"), (2, "VALID"), (3, "TRAIN"), (3, "VALID")]
    assert all(np.isfinite(h[2]) for h in r0["history"])


def test_rccl_overlapped_reduction_on_one_card():
    """RCCL itself, once (the scaling curve belongs to the driver's 8-GPU node): backend "nccl", world size 1,
    BucketedGradAllReduce forced active.  Three training steps go through the RCCL communicator -- the early bucket on the side
    stream under the haplotype-CNN / info-MLP backward, the late bucket and both stream joins at the optimizer step -- and must
    land where the un-hooked steps land.  (Two un-hooked runs differ in the last bits themselves: the small-parameter gradients
    are float atomics; the hooked run is held to that same run-to-run noise, with a floor of one fp32 ulp of the parameters.)"""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rccl.pt")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), out], cwd=ROOT, env=env,
                             capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        r = torch.load(out, weights_only=False)
    assert r["backend"] == "nccl" and r["early_reductions"] == 3 and r["side_stream"] and not r["pending"]
    assert 0 < r["late_start"] < r["plain"].numel()
    assert torch.isfinite(r["hooked"]).all()
    noise = float((r["plain"] - r["again"]).abs().max())
    diff = float((r["plain"] - r["hooked"]).abs().max())
    assert diff <= max(4 * noise, 2e-7 * float(r["plain"].abs().max())), (diff, noise)
    np.testing.assert_allclose(r["hooked_losses"], r["plain_losses"], rtol=1e-5)


def _bench_line(*flags, timeout=300):  # (well under the pool's 420 s silence limit: a hang must fail THIS test, not kill the run)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # ONE JSON line, from rank 0
    import json
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [(), ("--data", "loader", "--dataset-variants", "32768", "--chunk-variants", "16384"), ("--depth", "stress", "--batch", "128")],
                         ids=["resident", "loader", "stress"])
def test_bench_two_rank_path(extra):
    """bench.py's own N > 1 path (the one the driver launches on the 8-GPU node) must not run for the first time there: two ranks
    share this box's one card (--rehearse: gloo instead of RCCL, the numbers mean nothing), spawned the way `--gpus N` spawns them.
    The line must say two ranks took part, count both ranks' read sets, and show that the overlap hook fired once per step."""
    batch = "128" if "stress" in extra else "4096"
    flags = ["--gpus", "2", "--rehearse", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    if "--batch" not in extra:
        flags += ["--batch", batch]
    line = _bench_line(*flags, *extra)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    assert line["steps"] == 3 and line["warmup"] == 1
    per_rank = int(batch) * 3 / (line["ms_per_step"] * 3e-3)
    assert abs(line["value"] - 2 * per_rank) <= 1e-6 * line["value"]  # whole-job rate = both ranks' read sets / max-over-ranks time
    c = line["collective"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and c["op"] == "SUM"
    assert c["early_buckets_per_step"] == 1.0 and c["early_bytes"] > c["late_bytes"] > 0 and c["early_bucket_on_side_stream"]
    assert "cpu_baseline" not in line and line["filter"]["value"] > 0
