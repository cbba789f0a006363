"""The reference's on-disk dataset: a tar of four members (reference permutect/data/memory_mapped_data.py:198-285).

    metadata.metadata.npy        torch.save of uint32[5]: num_data, int_dim, float_dim, num_reads, reads_dim
    int_array.int_mmap.npy       np.save, int16   [>= num_data, 16 + H]
    float_array.float_mmap.npy   np.save, float16 [>= num_data, 6 + I]
    reads_array.reads_mmap.npy   np.save, uint8   [>= num_reads, 7 + nf]; per datum its ref rows, then its alt rows

(the arrays may carry unused capacity rows past num_data / num_reads: reference :41-42).  This module reads and writes
that format with the same class and method names, but keeps the data as plain (memory-mapped) arrays and CSR offsets:
nothing here builds per-variant Python objects, so a chunk can be shipped to HBM as it lies on disk
(`permutect_amd/data/reads_dataset.py`).  `generate()` still yields `Datum`s for code that wants them.
"""
from __future__ import annotations

import os
import tarfile
import tempfile
from typing import Generator, List, Optional

import numpy as np
import torch

from permutect_amd.data.datum import Data, Datum

SUFFIX_FOR_INT_MMAP = ".int_mmap.npy"
SUFFIX_FOR_FLOAT_MMAP = ".float_mmap.npy"
SUFFIX_FOR_READS_MMAP = ".reads_mmap.npy"
SUFFIX_FOR_METADATA = ".metadata.npy"


def _load_metadata(path):
    """The metadata member is a torch.save of a small numpy integer array (reference memory_mapped_data.py:208-219).
    A dataset tar is untrusted input: unpickle with the restricted loader, allowing only what such an array needs."""
    rec = np._core.multiarray._reconstruct
    allowed = [rec, (rec, "numpy.core.multiarray._reconstruct"), np.ndarray, np.dtype] + \
              [type(np.dtype(t)) for t in ("uint32", "int32", "int64", "uint64")]
    with torch.serialization.safe_globals(allowed):
        return torch.load(path, weights_only=True)


class MemoryMappedData:
    def __init__(self, int_mmap, float_mmap, num_data: int, reads_mmap, num_reads: int):
        self.int_mmap, self.float_mmap, self.reads_mmap = int_mmap, float_mmap, reads_mmap
        self.num_data, self.num_reads = int(num_data), int(num_reads)
        if self.reads_mmap is None:  # a dataset without reads (the posterior hand-off): no scan of a table that may be gigabytes
            per_datum = np.zeros(self.num_data, np.int64)
        else:
            counts = np.asarray(self.int_mmap[: self.num_data, : Data.ALT_COUNT.idx + 1]).astype(np.int64)
            per_datum = counts[:, Data.REF_COUNT.idx] + counts[:, Data.ALT_COUNT.idx] if self.num_data else np.zeros(0, np.int64)
        # reference :51-55 builds this with a Python loop over Datums; same values (uint32, like the reference)
        self.read_end_indices = np.cumsum(per_datum).astype(np.uint32)
        if self.reads_mmap is not None and self.num_data:
            assert int(self.read_end_indices[-1]) <= self.num_reads, "read counts exceed the reads array"
        self._temp_dir = None

    def __len__(self) -> int:
        return self.num_data

    def num_bytes(self) -> int:
        return self.int_mmap.nbytes + self.float_mmap.nbytes + (0 if self.reads_mmap is None else self.reads_mmap.nbytes)

    def read_start_indices(self) -> np.ndarray:
        """int64 [num_data + 1]: CSR row offsets of every datum's reads (ref rows first)."""
        out = np.zeros(self.num_data + 1, dtype=np.int64)
        out[1:] = self.read_end_indices
        return out

    @classmethod
    def from_arrays(cls, int_array: np.ndarray, float_array: np.ndarray, reads: Optional[np.ndarray]) -> "MemoryMappedData":
        return cls(np.ascontiguousarray(int_array, dtype=np.int16), np.ascontiguousarray(float_array, dtype=np.float16),
                   len(int_array), None if reads is None else np.ascontiguousarray(reads), 0 if reads is None else len(reads))

    # ---- reference-compatible iteration (reference :64-88) -------------------------------------------------------------
    def generate(self, num_folds: int = 1, used_folds: List[int] = None) -> Generator[Datum, None, None]:
        folds = None if used_folds is None else set(used_folds)
        starts = self.read_start_indices()
        for idx in range(self.num_data):
            if folds is None or (idx % num_folds) in folds:
                reads = (np.zeros((0, 0), dtype=np.uint8) if self.reads_mmap is None
                         else self.reads_mmap[starts[idx]:starts[idx + 1]])
                yield Datum(self.int_mmap[idx], self.float_mmap[idx], reads, compressed=reads.dtype == np.uint8)

    def fold_indices(self, num_folds: int, used_folds: Optional[List[int]]) -> np.ndarray:
        """Indices of the data in the given folds (`idx % num_folds in used_folds`, reference :75)."""
        idx = np.arange(self.num_data, dtype=np.int64)
        if used_folds is None:
            return idx
        return idx[np.isin(idx % num_folds, np.asarray(sorted(set(used_folds)), dtype=np.int64))]

    def restrict_to_folds(self, num_folds: int, used_folds: List[int] = None) -> "MemoryMappedData":
        """A new dataset holding only the given folds, in order (reference :90-106; a vectorised copy here)."""
        if used_folds is None:
            return self
        keep = self.fold_indices(num_folds, used_folds)
        starts = self.read_start_indices()
        if self.reads_mmap is None:
            reads = None
        else:
            lengths = (starts[keep + 1] - starts[keep]).astype(np.int64)
            out_start = np.zeros(len(keep) + 1, dtype=np.int64)
            np.cumsum(lengths, out=out_start[1:])
            rows = np.repeat(starts[keep] - out_start[:-1], lengths) + np.arange(int(out_start[-1]), dtype=np.int64)
            reads = np.asarray(self.reads_mmap)[rows]
        return MemoryMappedData.from_arrays(np.asarray(self.int_mmap[: self.num_data])[keep],
                                            np.asarray(self.float_mmap[: self.num_data])[keep], reads)

    # ---- tar format (reference :198-285) ---------------------------------------------------------------------------------
    def save_to_tarfile(self, output_tarfile):
        metadata = np.array([self.num_data, self.int_mmap.shape[-1], self.float_mmap.shape[-1], self.num_reads,
                             0 if self.reads_mmap is None else self.reads_mmap.shape[-1]], dtype=np.uint32)
        with tempfile.TemporaryDirectory() as tmp:
            members = []

            def add(name, writer):
                path = os.path.join(tmp, name)
                writer(path)
                members.append((path, name))

            add("metadata" + SUFFIX_FOR_METADATA, lambda p: torch.save(metadata, p))
            add("int_array" + SUFFIX_FOR_INT_MMAP, lambda p: np.save(p, np.asarray(self.int_mmap[: self.num_data])))
            add("float_array" + SUFFIX_FOR_FLOAT_MMAP, lambda p: np.save(p, np.asarray(self.float_mmap[: self.num_data])))
            if self.reads_mmap is not None:
                add("reads_array" + SUFFIX_FOR_READS_MMAP, lambda p: np.save(p, np.asarray(self.reads_mmap[: self.num_reads])))
            with tarfile.open(output_tarfile, "w") as tar:
                for path, name in members:
                    tar.add(path, arcname=name)

    @classmethod
    def load_from_tarfile(cls, data_tarfile) -> "MemoryMappedData":
        temp_dir = tempfile.TemporaryDirectory()
        with tarfile.open(data_tarfile, "r") as tar:
            for member in tar.getmembers():
                if member.isfile():
                    member.name = os.path.basename(member.name)  # flat extraction, nothing escapes the temp dir
                    tar.extract(member, path=temp_dir.name)
        files = [os.path.join(temp_dir.name, p) for p in os.listdir(temp_dir.name)]

        def one(suffix, required=True):
            found = [f for f in files if f.endswith(suffix)]
            assert len(found) == 1 or (not required and not found), f"expected one *{suffix} member, found {len(found)}"
            return found[0] if found else None

        metadata = _load_metadata(one(SUFFIX_FOR_METADATA))
        num_data, int_dim, float_dim, num_reads, reads_dim = (int(x) for x in np.asarray(metadata)[:5])
        int_mmap = np.load(one(SUFFIX_FOR_INT_MMAP), mmap_mode="r")
        float_mmap = np.load(one(SUFFIX_FOR_FLOAT_MMAP), mmap_mode="r")
        reads_file = one(SUFFIX_FOR_READS_MMAP, required=num_reads > 0)
        reads_mmap = None if (num_reads == 0 or reads_file is None) else np.load(reads_file, mmap_mode="r")
        assert int_mmap.shape[-1] == int_dim and float_mmap.shape[-1] == float_dim and len(int_mmap) >= num_data
        assert reads_mmap is None or (reads_mmap.shape[-1] == reads_dim and len(reads_mmap) >= num_reads)
        result = cls(int_mmap, float_mmap, num_data, reads_mmap, num_reads)
        result._temp_dir = temp_dir  # keeps the extracted files alive while the memory maps are in use
        return result
