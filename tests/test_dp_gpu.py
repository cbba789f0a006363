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


def test_two_rank_training_keeps_replicas_identical():
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", "29631", os.path.join(ROOT, "tests", "dp_worker.py"), d]
        res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
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
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29647")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), out], cwd=ROOT, env=env,
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        r = torch.load(out, weights_only=False)
    assert r["backend"] == "nccl" and r["early_reductions"] == 3 and r["side_stream"] and not r["pending"]
    assert 0 < r["late_start"] < r["plain"].numel()
    assert torch.isfinite(r["hooked"]).all()
    noise = float((r["plain"] - r["again"]).abs().max())
    diff = float((r["plain"] - r["hooked"]).abs().max())
    assert diff <= max(4 * noise, 2e-7 * float(r["plain"].abs().max())), (diff, noise)
    np.testing.assert_allclose(r["hooked_losses"], r["plain_losses"], rtol=1e-5)
