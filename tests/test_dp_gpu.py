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


def test_bench_two_rank_sharded_filter_pass():
    """bench.py's N > 1 line also carries the filter as it is split over GPUs: one dataset, a contiguous shard per rank, the posterior
    rows concatenated on rank 0 (`filter_sharded_dataset`)."""
    line = _bench_line("--gpus", "2", "--rehearse", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--batch", "4096")
    f = line["filter_sharded_dataset"]
    assert line["n_gpus"] == 2 and f["rows_on_rank0"] == f["candidates"] > 0 and f["value"] > 0


def _torchrun(module, args, nproc, timeout=300):
    """`python -m torch.distributed.run --nproc-per-node N -m <module> ...` the way a user launches the tools; N ranks share this box's
    one card (PMT_DIST_BACKEND=gloo: RCCL wants a device per rank)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", PMT_DIST_BACKEND="gloo", PMT_JIT="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), "-m", module, *args]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    return res.stdout


def test_filter_tool_two_ranks_writes_the_single_process_rows():
    """`filter_variants` under torchrun -- the candidates in two contiguous shards, one process each through the HIP
    forward, rank 0 concatenating in dataset order -- writes the posterior tar of the single-process run."""
    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.data.memory_mapped_data import MemoryMappedData
    from permutect_amd.parameters import P0_DIMS, p0_params
    tar = os.path.join(ROOT, "tests", "golden", "tiny_dataset.tar")
    with tempfile.TemporaryDirectory() as d:
        torch.manual_seed(11)
        model = ArtifactModel(p0_params(), device=torch.device("cuda"), **P0_DIMS)
        model.save_model(os.path.join(d, "model.pt"))
        common = ["--test_dataset_tar", tar, "--artifact_model", os.path.join(d, "model.pt"), "--batch_size", "16"]
        env = dict(os.environ, PMT_JIT="0")
        one = subprocess.run([sys.executable, "-m", "permutect_amd.tools.filter_variants", *common, "--output", os.path.join(d, "one.tar")],
                             cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
        out = _torchrun("permutect_amd.tools.filter_variants", [*common, "--output", os.path.join(d, "two.tar")], 2)
        assert "on 2 GPU(s)" in out
        a, b = (MemoryMappedData.load_from_tarfile(os.path.join(d, f)) for f in ("one.tar", "two.tar"))
        n = len(a)
        assert n == len(b) and n > 20
        assert np.array_equal(np.asarray(a.int_mmap[:n]), np.asarray(b.int_mmap[:n]))
        # the rows' PLACES are exact (integer rows bit for bit; tests/test_distributed_cpu.py shows the concatenation itself bit-identical
        # with a deterministic forward); the HIP forward's per-set sums are float atomics whose order changes with the batch a variant
        # sits in, so its numbers agree to the last bits only: the logit through its float16 store, the embedding to 1e-5
        fa, fb = np.asarray(a.float_mmap[:n]).astype(np.float32), np.asarray(b.float_mmap[:n]).astype(np.float32)
        assert fa.shape == fb.shape and np.isfinite(fa).all() and np.abs(fa[:, 6:]).max() > 0  # embeddings, not zeros
        np.testing.assert_allclose(fb[:, :6], fa[:, :6], rtol=1e-3, atol=1e-3)   # (one float16 ulp of a logit of ~12)
        np.testing.assert_allclose(fb[:, 6:], fa[:, 6:], rtol=1e-5, atol=1e-5)


def test_train_tool_two_ranks_keeps_replicas_identical():
    """`train_artifact_model` under torchrun: the tool itself initialises the process group (training/distributed.py: init_from_env),
    trains data parallel and ends with the replica check (assert_replicas_identical raises -- a non-zero exit -- when the ranks'
    parameters differ in a single bit); rank 0 writes a loadable model."""
    from permutect_amd.architecture.artifact_model import load_model
    tar = os.path.join(ROOT, "tests", "golden", "tiny_dataset.tar")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "model.pt")
        args = ["--train_tar", tar, "--output", out, "--read_layers", "30", "-2", "--info_layers", "20", "-2", "--aggregation_layers", "-2", "10",
                "--self_attention_hidden_dimension", "20", "--num_self_attention_layers", "2", "--num_artifact_clusters", "4",
                "--calibration_layers", "10", "10", "--ref_seq_layer_strings", "convolution/kernel_size=3/out_channels=32", "pool/kernel_size=2",
                "leaky_relu", "convolution/kernel_size=3/out_channels=32", "leaky_relu", "flatten", "linear/out_features=10", "--dropout_p", "0.0", "--batch_size", "8", "--num_epochs", "2",
                "--num_calibration_epochs", "1", "--inference_batch_size", "16", "--learning_rate", "0.001"]
        stdout = _torchrun("permutect_amd.tools.train_artifact_model", args, 2, timeout=600)
        assert "epoch 2 TRAIN" in stdout and stdout.count("epoch 1 TRAIN") == 1  # rank 0 alone reports
        model, _, _ = load_model(out, device=torch.device("cuda"))
        assert all(torch.isfinite(p).all() for p in model.parameters())
