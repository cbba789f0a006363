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
