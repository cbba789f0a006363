"""CPU, gloo, world_size 2: the data-parallel semantics of SURVEY.md 8e.  Compute is the CPU oracle (tests may use
it); what is under test is the sharding + single flat SUM all-reduce + identical clip/AdamW on every rank."""
import os
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import artifact_oracle as O
from permutect_amd.training.distributed import BucketedGradAllReduce, GradAllReduce, rank0_decides, shard_range
from tests.helpers import config_for, load_case


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _sub_batch(b, lo, hi):
    nref, nalt = b["nref"].numpy(), b["nalt"].numpy()
    ro, ao = np.concatenate([[0], np.cumsum(nref)]), np.concatenate([[0], np.cumsum(nalt)]) + nref.sum()
    rows = np.concatenate([np.arange(ro[lo], ro[hi]), np.arange(ao[lo], ao[hi])])
    out = {k: b[k][lo:hi] for k in ("nref", "nalt", "labels", "sources", "info_be", "haplotypes_bh")}
    out["reads_re"] = b["reads_re"][torch.from_numpy(rows)]
    return out


def _worker(rank, world, init_file, result_file):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(1)
    z, sd, b = load_case("p0_b16")
    cfg = config_for("p0_b16")
    lo, hi = shard_range(16, rank, world)
    _, _, grads = O.train_step_grads(sd, cfg, _sub_batch(b, lo, hi))
    names = sorted(grads)
    flat = torch.cat([grads[n].reshape(-1) for n in names])
    GradAllReduce()(flat)  # ONE collective over the flat buffer
    params = [sd[n].clone() for n in names]
    m, v = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    off, gl = 0, []
    for p in params:
        gl.append(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    norm = O.clip_and_adamw(params, gl, m, v, step=1, lr=1e-3, weight_decay=0.01)
    decision = rank0_decides(rank == 0, torch.device("cpu"))
    torch.save({"flat": flat, "params": torch.cat([p.reshape(-1) for p in params]), "norm": norm, "decision": decision},
               f"{result_file}.{rank}")
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_full_batch():
    z, sd, b = load_case("p0_b16")
    names = sorted(k[5:] for k in z.files if k.startswith("grad/"))
    ref_flat = np.concatenate([z["grad/" + n].ravel() for n in names])  # the reference's gradient on the FULL batch
    with tempfile.TemporaryDirectory() as d:
        init_file, result_file = os.path.join(d, "init"), os.path.join(d, "res")
        mp.spawn(_worker, args=(2, init_file, result_file), nprocs=2, join=True)
        r0, r1 = torch.load(result_file + ".0"), torch.load(result_file + ".1")
    # both ranks hold the same reduced gradient, equal to the full-batch gradient (loss is a SUM over the batch)
    assert torch.equal(r0["flat"], r1["flat"])
    assert np.linalg.norm(r0["flat"].numpy() - ref_flat) <= 1e-4 * np.linalg.norm(ref_flat)
    # ... and took the identical optimizer step
    assert torch.equal(r0["params"], r1["params"]) and r0["norm"] == r1["norm"]
    ref_after = np.concatenate([z["after/" + n].ravel() for n in names])
    assert np.abs(r0["params"].numpy() - ref_after).max() <= 0.02 * 1e-3
    assert r0["decision"] is True and r1["decision"] is True  # rank 0's flag wins everywhere


def _bucket_worker(rank, world, init_file, result_file):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(59848, generator=g)
    want = flat.clone()
    GradAllReduce()(want)                                   # the flat SUM all-reduce
    out = {}
    for late_start in (0, 1, 55000, 59848):                 # empty early bucket, tiny, the real split, empty late bucket
        hook = BucketedGradAllReduce()
        got = flat.clone()
        hook.early(got, late_start)                         # (what ReadSetEngine.backward does after pmt_backward)
        got[late_start:] += 0.0                             # the late kernels still write behind the early bucket
        hook(got)                                           # (what FusedClipAdamW.step(pre_reduce=hook) does)
        out[late_start] = torch.equal(got, want)
    hook = BucketedGradAllReduce()
    got = flat.clone()
    hook(got)                                               # no early() since the last step: one flat reduction
    out["flat"] = torch.equal(got, want)
    got2 = flat.clone()
    hook.early(got2, 55000)
    try:
        hook.early(got2, 55000)                             # a second backward before the step: refused, never a silent wrong sum
        out["twice"] = False
    except RuntimeError:
        out["twice"] = True
    hook(got2)
    out["after_refusal"] = torch.equal(got2, want)
    torch.save(out, f"{result_file}.{rank}")
    dist.destroy_process_group()


def test_bucketed_all_reduce_equals_flat_sum():
    """The overlapped two-bucket hook's arithmetic: early bucket + late bucket == one flat SUM all-reduce, bit for bit
    (a SUM all-reduce of a slice is the slice of the SUM all-reduce), for every split point and on both ranks."""
    with tempfile.TemporaryDirectory() as d:
        init_file, result_file = os.path.join(d, "init"), os.path.join(d, "res")
        mp.spawn(_bucket_worker, args=(2, init_file, result_file), nprocs=2, join=True)
        for r in range(2):
            res = torch.load(result_file + f".{r}")
            assert all(res.values()), res


class _FakeArtifactModel:
    """What make_posterior_mmap asks of a model, with a forward that is a pure function of each variant's OWN rows (torch on the CPU):
    the rows it produces cannot depend on how the candidates were cut into shards, chunks or batches -- which is the property under
    test.  (The real forward is the HIP kernels: tests/test_dp_gpu.py runs the same comparison through them.)"""

    class _Reducer:
        @staticmethod
        def output_dimension():
            return 10

    class _Engine:
        @staticmethod
        def check_join_fault():
            return None

    class _Out:
        pass

    def __init__(self):
        self._device, self.reducer = torch.device("cpu"), self._Reducer()

    def train(self, mode):
        return self

    def engine(self):
        return self._Engine()

    def compute_batch_output(self, batch):
        from permutect_amd.data.datum import INFO_START_IDX
        out = self._Out()
        info = batch.float_tensor[:, INFO_START_IDX:].float()
        out.logits_b = info.sum(dim=1) * 0.37 + batch.int_tensor[:, 0].float() - 0.5 * batch.int_tensor[:, 1].float()
        out.features_be = torch.tanh(info[:, :10]) * 3.0 + batch.int_tensor[:, 16:26].float()
        return out


def _posterior_worker(rank, world, init_file, result_file, batch_size, chunk_variants):
    from permutect_amd.data.memory_mapped_data import MemoryMappedData
    from permutect_amd.data.reads_dataset import ReadsDataset
    from permutect_amd.tools.posterior_data import make_posterior_mmap
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(1)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ds = ReadsDataset(MemoryMappedData.load_from_tarfile(os.path.join(root, "tests", "golden", "tiny_dataset.tar")))
    res = make_posterior_mmap(ds, _FakeArtifactModel(), batch_size, device=torch.device("cpu"), chunk_variants=chunk_variants, rank=rank, world_size=world)
    assert (res is None) == (rank != 0)
    if rank == 0:
        np.savez(result_file, ints=np.asarray(res.int_mmap[: len(res)]), floats=np.asarray(res.float_mmap[: len(res)]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_posterior_rows_are_bit_identical_to_the_single_process_ones():
    """SURVEY 8e, the filter: the candidates cut into contiguous shards, one rank each, no collective on the data path, the rows
    concatenated on rank 0 in dataset order (tools/posterior_data.py: make_posterior_mmap(..., rank, world_size)) -- against the same
    function in ONE process, for two and three ranks, batches that do and do not divide the shards."""
    from permutect_amd.data.memory_mapped_data import MemoryMappedData
    from permutect_amd.data.reads_dataset import ReadsDataset
    from permutect_amd.tools.posterior_data import make_posterior_mmap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ds = ReadsDataset(MemoryMappedData.load_from_tarfile(os.path.join(root, "tests", "golden", "tiny_dataset.tar")))
    single = make_posterior_mmap(ds, _FakeArtifactModel(), 7, device=torch.device("cpu"))
    n = len(single)
    assert n == len(ds) and n > 20
    for world, batch_size, chunk in ((2, 7, None), (3, 5, 11)):
        with tempfile.TemporaryDirectory() as d:
            init_file, result_file = os.path.join(d, "init"), os.path.join(d, "res.npz")
            mp.spawn(_posterior_worker, args=(world, init_file, result_file, batch_size, chunk), nprocs=world, join=True)
            z = np.load(result_file)
            assert z["ints"].shape[0] == n
            assert np.array_equal(z["ints"], np.asarray(single.int_mmap[:n]))
            assert np.array_equal(z["floats"].view(np.uint32), np.asarray(single.float_mmap[:n]).view(np.uint32))  # bit for bit


def test_init_from_env_single_process_is_a_no_op(monkeypatch):
    from permutect_amd.training.distributed import init_from_env
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    d, rank, world, device = init_from_env(device_type="cpu")
    assert d is None and (rank, world) == (0, 1) and device.type == "cpu"
