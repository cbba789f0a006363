"""The on-disk dataset format and the host collate against a tar written by the reference itself
(tests/golden/tiny_dataset.tar + tiny_dataset_expected.npz, made by tests/golden/make_golden.py --dataset-only)."""
import os

import numpy as np

from permutect_amd.data.batch import decode_packed_reads
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.data.reads_dataset import ReadsDataset

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _expected():
    return np.load(os.path.join(GOLDEN, "tiny_dataset_expected.npz"))


def test_reads_reference_tar():
    z = _expected()
    mm = MemoryMappedData.load_from_tarfile(os.path.join(GOLDEN, "tiny_dataset.tar"))
    assert mm.num_data == int(z["num_data"]) and mm.num_reads == int(z["num_reads"])
    assert len(mm.int_mmap) > mm.num_data  # the reference's file carries unused capacity rows: they must be ignored
    np.testing.assert_array_equal(mm.read_end_indices, z["read_end_indices"])
    np.testing.assert_array_equal(np.asarray(mm.int_mmap[: mm.num_data]), z["int_array"])
    np.testing.assert_array_equal(np.asarray(mm.float_mmap[: mm.num_data]), z["float_array"])
    np.testing.assert_array_equal(np.asarray(mm.reads_mmap[: mm.num_reads]), z["reads"])
    datums = list(mm.generate())
    assert len(datums) == mm.num_data and datums[5].reads_re.shape == (1, 12)


def test_tar_round_trip(tmp_path):
    z = _expected()
    mm = MemoryMappedData.from_arrays(z["int_array"], z["float_array"], z["reads"])
    path = tmp_path / "out.tar"
    mm.save_to_tarfile(str(path))
    back = MemoryMappedData.load_from_tarfile(str(path))
    assert back.num_data == mm.num_data and back.num_reads == mm.num_reads
    np.testing.assert_array_equal(np.asarray(back.reads_mmap), z["reads"])
    np.testing.assert_array_equal(np.asarray(back.float_mmap), z["float_array"])
    import tarfile
    with tarfile.open(path) as tar:  # the member names the reference's loader looks for (memory_mapped_data.py:246-249)
        names = sorted(m.name for m in tar.getmembers())
    assert names == ["float_array.float_mmap.npy", "int_array.int_mmap.npy", "metadata.metadata.npy", "reads_array.reads_mmap.npy"]


def test_fold_split_matches_reference():
    z = _expected()
    mm = MemoryMappedData.load_from_tarfile(os.path.join(GOLDEN, "tiny_dataset.tar"))
    np.testing.assert_array_equal(mm.fold_indices(3, [0, 2]), z["fold_ids"])
    r = mm.restrict_to_folds(3, [0, 2])
    assert r.num_data == int(z["fold_num_data"]) and r.num_reads == int(z["fold_num_reads"])
    np.testing.assert_array_equal(np.asarray(r.int_mmap), z["fold_int"])
    np.testing.assert_array_equal(np.asarray(r.reads_mmap), z["fold_reads"])


def test_host_collate_matches_reference_batch():
    z = _expected()
    ds = ReadsDataset(MemoryMappedData.load_from_tarfile(os.path.join(GOLDEN, "tiny_dataset.tar")))
    assert (ds.num_read_features(), ds.num_info_features(), ds.haplotypes_length(), ds.num_sources()) == (61, 71, 42, 2)
    b = ds.host_batch(z["batch_ids"])
    np.testing.assert_array_equal(b.int_tensor.numpy(), z["batch_int"])
    np.testing.assert_array_equal(b.float_tensor.numpy(), z["batch_float"])
    # ref rows of every variant, then alt rows, decoded with the reference's uint8 wrap rule
    np.testing.assert_array_equal(decode_packed_reads(b.packed_reads.numpy()), z["batch_reads_f16"])
    assert b.plan().num_groups >= 1


def test_loader_covers_every_variant_once():
    ds = ReadsDataset(MemoryMappedData.load_from_tarfile(os.path.join(GOLDEN, "tiny_dataset.tar")))
    seen = []
    loader = ds.make_data_loader(batch_size=8, chunk_variants=16, rng=np.random.default_rng(3))
    for b in loader:
        assert b.size() <= 8
        seen.append(b.int_tensor[:, 16:].numpy())
    assert len(loader) == len(seen)
    got = np.concatenate(seen)
    want = np.asarray(ds._ints[: len(ds), 16:]).astype(np.int64)
    assert got.shape == want.shape
    assert sorted(map(bytes, got)) == sorted(map(bytes, want))


def test_posterior_rows_match_the_reference_per_datum_loop():
    """tools/posterior_data.posterior_rows against what the REFERENCE's loop in generate_posterior_data
    (tools/filter_variants.py:302-320) made of the same batch (tests/golden/posterior_rows.npz, written by make_golden.py
    --posterior-only): counts zeroed, the logit stored through the float16 scalar array, info := embedding.  Integers bit for
    bit; floats exactly too (the logit's float16 rounding included)."""
    import torch
    from permutect_amd.tools.posterior_data import posterior_rows
    z = np.load(os.path.join(GOLDEN, "posterior_rows.npz"))
    ints, floats = posterior_rows(torch.from_numpy(z["batch_int"]), torch.from_numpy(z["batch_float"]),
                                  torch.from_numpy(z["logits_b"]), torch.from_numpy(z["features_be"]))
    assert ints.dtype == torch.int16 and str(z["out_float_dtype"]) == "float32" and floats.dtype == torch.float32
    np.testing.assert_array_equal(ints.numpy(), z["out_int"])
    np.testing.assert_array_equal(floats.numpy(), z["out_float"])
    # saturated and tiny logits survive the float16 hop as the reference's do
    assert floats[0, 5] == 20.0 and floats[1, 5] == -20.0 and float(floats[2, 5]) == float(np.float16(1e-4))



def test_validate_sources_requires_data_for_every_source():
    """reference data/reads_dataset.py:212-221"""
    import pytest
    from permutect_amd.data.datum import Data
    z = _expected()
    ints = z["int_array"].copy()
    ints[:, Data.SOURCE.idx] = 0
    ds = ReadsDataset(MemoryMappedData.from_arrays(ints, z["float_array"], z["reads"]))
    assert ds.validate_sources() == 1
    ints[::2, Data.SOURCE.idx] = 2  # sources 0 and 2 present, 1 missing
    ds = ReadsDataset(MemoryMappedData.from_arrays(ints, z["float_array"], z["reads"]))
    assert ds.num_sources() == 3
    with pytest.raises(AssertionError, match="No data for source 1"):
        ds.validate_sources()
    ints[1::4, Data.SOURCE.idx] = 1
    assert ReadsDataset(MemoryMappedData.from_arrays(ints, z["float_array"], z["reads"])).validate_sources() == 3


def test_prepare_chunk_matches_the_python_path():
    """pmt_prepare_chunk (one GIL-free call per chunk: counts, consumption order, every batch's group plan) against the loader's
    Python path (`_prepare`: numpy gathers + pmt_pack_order_batches + one GroupPlan per batch): same ids, same plans, batch by
    batch, without shuffling; with shuffling a permutation of the chunk whose batches carry exactly the plans of their own counts."""
    import torch
    from bench import synth_arrays
    from permutect_amd.data.batch import GroupPlan
    from permutect_amd.data.reads_dataset import DeviceChunk, DeviceChunkLoader, PinnedStage
    ints, floats, packed = synth_arrays(np.random.default_rng(8), 3000, "wgs")
    ds = ReadsDataset(MemoryMappedData.from_arrays(ints, floats, packed))
    cpu = torch.device("cpu")
    for shuffle in (False, True):
        loader = DeviceChunkLoader(ds, 512, cpu, None, np.random.default_rng(1), shuffle, 0, 1)
        chunk = DeviceChunk(ds, 100, 2900, cpu)
        fast = loader._prepare_fast(chunk, seed=77, stage=PinnedStage())
        assert fast is not None and len(fast) == 6  # 2800 variants: five batches of 512 and one of 240
        rc, ac = chunk.host_counts()
        assert np.array_equal(rc, ints[100:2900, 0]) and np.array_equal(ac, ints[100:2900, 1])
        all_ids = np.concatenate([ids for ids, _, _ in fast])
        assert np.array_equal(np.sort(all_ids), np.arange(2800))
        if not shuffle:
            slow = loader._prepare(DeviceChunk(ds, 100, 2900, cpu), np.arange(2800), PinnedStage())
            for (ids_f, dev_f, plan_f), (ids_s, dev_s, plan_s) in zip(fast, slow):
                assert np.array_equal(ids_f, ids_s) and np.array_equal(dev_f.numpy(), ids_s)
                assert np.array_equal(plan_f.group_start, plan_s.group_start) and np.array_equal(plan_f.group_tile_base, plan_s.group_tile_base)
                assert plan_f.num_groups == plan_s.num_groups and plan_f.total_tiles == plan_s.total_tiles
        else:
            assert not np.array_equal(all_ids, np.arange(2800))
        for ids_f, dev_f, plan_f in fast:
            want = GroupPlan(rc[ids_f], ac[ids_f])
            assert np.array_equal(plan_f.group_start, want.group_start) and np.array_equal(plan_f.group_tile_base, want.group_tile_base)
            gs, gt, span = plan_f.on(cpu)
            assert span is None and np.array_equal(gs.numpy(), want.group_start) and np.array_equal(gt.numpy(), want.group_tile_base)
            assert plan_f.total_reads == int(rc[ids_f].sum() + ac[ids_f].sum())
    # a read set beyond one workgroup: the fast path declines, the loader plans that chunk with the split planner
    big = ints.copy()
    big[5, 0] = 300
    reads = np.zeros((int(big[:, 0].astype(np.int64).sum() + big[:, 1].astype(np.int64).sum()), 12), dtype=np.uint8)
    ds2 = ReadsDataset(MemoryMappedData.from_arrays(big, floats, reads))
    loader = DeviceChunkLoader(ds2, 512, cpu, None, np.random.default_rng(1), False, 0, 1)
    assert loader._prepare_fast(DeviceChunk(ds2, 0, 3000, cpu), seed=1, stage=PinnedStage()) is None
    assert sum(b.size() for b in loader) == 3000
