"""Dataset + loaders over `MemoryMappedData` (reference permutect/data/reads_dataset.py:47-232).

The reference iterates Datum by Datum (chunks of the memory map in shuffled order, shuffled within a chunk, reference
:141-196) and collates every batch in Python (`Batch(List[Datum])`, 0.27 s per 8192 variants) before uploading
244 B per read.  Here the unit of host work is the CHUNK:

  * `device_loader`: a chunk (a contiguous range of the dataset, as it lies on disk: packed 12-byte read rows, int16 /
    float16 per-variant rows, CSR offsets) is uploaded to HBM once; every batch is then composed ON THE DEVICE from a
    shuffled slice of the chunk's variant ids: two row gathers for the per-variant arrays and `pmt_build_read_index`
    for the reads, which the kernels consume through their gather index without moving a read row.
  * `make_data_loader`: host batches (same `Batch` objects, packed reads) with a vectorised numpy collate, for callers
    that want the reference's DataLoader shape.

Same sampling scheme as the reference (shuffled chunks, shuffled variants within a chunk, fold split `idx % num_folds`);
batches do not straddle chunks (the reference's DataLoader lets the last batch of a chunk continue into the next one).
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional

import numpy as np
import torch

from permutect_amd.data.batch import Batch, GroupPlan
from permutect_amd.data.datum import (Data, Datum, HAPLOTYPES_START_IDX, INFO_START_IDX,
                                      NUMBER_OF_BYTES_IN_PACKED_READ)
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.engine import lib as L


def last_fold_only(num_folds: int):
    return [num_folds - 1]


def all_but_last_fold(num_folds: int):
    return list(range(num_folds - 1))


def all_but_one_fold(num_folds: int, fold_to_exclude: int):
    return list(range(fold_to_exclude)) + list(range(fold_to_exclude + 1, num_folds))


def all_folds(num_folds: int):
    return list(range(num_folds))


def _gather_rows(starts: np.ndarray, ids: np.ndarray, skip: np.ndarray, lengths: np.ndarray) -> np.ndarray:
    """Row indices `starts[ids] + skip + [0, lengths)` concatenated over ids (vectorised)."""
    total = int(lengths.sum())
    out_start = np.zeros(len(ids) + 1, dtype=np.int64)
    np.cumsum(lengths, out=out_start[1:])
    return np.repeat(starts[ids] + skip - out_start[:-1], lengths) + np.arange(total, dtype=np.int64)


class ReadsDataset:
    def __init__(self, memory_mapped_data: MemoryMappedData, num_folds: int = 1, folds_to_use: List[int] = None):
        self.memory_mapped_data = memory_mapped_data.restrict_to_folds(num_folds, folds_to_use)
        d = self.memory_mapped_data
        self._size = d.num_data
        self._ints = d.int_mmap
        self._floats = d.float_mmap
        self._reads = d.reads_mmap
        self._starts = d.read_start_indices()
        labels = np.asarray(self._ints[: self._size, Data.LABEL.idx]).astype(np.int64)
        self.totals_by_label_l = torch.from_numpy(np.bincount(labels, minlength=3).astype(np.float32))
        sources = np.asarray(self._ints[: self._size, Data.SOURCE.idx]).astype(np.int64)
        self._num_sources = int(sources.max()) + 1 if self._size else 1
        self._totals_slvra = None
        self._pinned = None  # pin_memory(): the arrays as page-locked torch tensors (ints, floats, reads, starts)

    def pin_memory(self) -> "ReadsDataset":
        """Move the dataset's arrays into page-locked host memory (once; a dataset that fits host memory).  The device loader
        then sends a chunk to HBM by DMA straight from the dataset -- no staging copy, which at ~25 GB/s on six threads is
        slower than the filter forward consumes data (28 MB per 65 536-variant batch every 0.9 ms = 30 GB/s).  A dataset that
        stays a memory map of a file (larger than memory) keeps the staged path."""
        if self._pinned is None and torch.cuda.is_available():
            def pin(arr):
                arr = np.asarray(arr)
                t = torch.empty(arr.shape, dtype=torch.from_numpy(np.empty(0, arr.dtype)).dtype, pin_memory=True)
                if arr.flags["C_CONTIGUOUS"] and arr.nbytes >= (1 << 22):
                    L.check(L.load().pmt_host_copy(t.data_ptr(), arr.ctypes.data, arr.nbytes, _STAGE_THREADS), "pmt_host_copy")
                else:
                    np.copyto(t.numpy(), arr)
                return t
            nreads = int(self._starts[self._size])
            ints, floats = pin(self._ints[: self._size]), pin(self._floats[: self._size])
            reads = pin(self._reads[:nreads]) if self._reads is not None else None
            self._pinned = (ints, floats, reads, pin(self._starts))
            self._ints, self._floats = ints.numpy(), floats.numpy()
            self._reads = None if reads is None else reads.numpy()
        return self

    def host_bytes(self) -> int:
        nreads = int(self._starts[self._size])
        per_read = self._reads.shape[1] * self._reads.itemsize if self._reads is not None else 0
        return int(self._ints[: self._size].nbytes + self._floats[: self._size].nbytes + nreads * per_read + np.asarray(self._starts).nbytes)

    def pin_memory_if_it_fits(self, max_fraction: float = 0.25) -> bool:
        """What the tools call before they stream a dataset (train_artifact_model, make_posterior_mmap): page-lock it (pin_memory)
        when it takes at most this process's SHARE of `max_fraction` of the host memory that is available right now; otherwise (or
        with PMT_PIN_DATASET=0 in the environment) the dataset stays where it is -- a memory map of a file, typically -- and chunks
        go through the staging copies.  The share: every rank of a data-parallel job on this node (LOCAL_WORLD_SIZE, set by torchrun)
        makes the same decision at the same moment and page-locks its OWN copy, so the budget is divided by their number (ADVICE r4:
        eight ranks each taking 40 % of "available" is three times the machine); pinned memory cannot be swapped, hence a quarter,
        not more.  A pin that fails all the same (hipHostMalloc refused: another process got there first) leaves the staged path.
        Returns whether the dataset is page-locked."""
        if self._pinned is not None:
            return True
        if os.environ.get("PMT_PIN_DATASET", "1") == "0" or not torch.cuda.is_available():
            return False
        try:
            import psutil
            available = psutil.virtual_memory().available
        except Exception:  # noqa: BLE001 -- no way to tell: leave the dataset alone
            return False
        try:
            local_ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        except ValueError:
            local_ranks = 1
        if self.host_bytes() > max_fraction * available / local_ranks:
            return False
        try:
            self.pin_memory()
        except (RuntimeError, MemoryError) as exc:  # page-locking refused: nothing was replaced (pin_memory assigns at its end)
            import warnings
            warnings.warn(f"permutect_amd: page-locking the dataset failed ({str(exc)[:120]}); chunks go through the staging copies")
            self._pinned = None
            return False
        return self._pinned is not None

    # ---- reference accessors (reference :95-112, :198-221) ---------------------------------------------------------------
    def totals_by_label(self):
        return self.totals_by_label_l

    @property
    def totals_slvra(self) -> torch.Tensor:
        """Number of data per (source, label, variant type, ref-count bin, alt-count bin): the reference's
        `totals_slvra` (reads_dataset.py:84-88 records every datum), one vectorised pass here.  Input of the downsampler's
        balance fit (training/model_training.py:60)."""
        if self._totals_slvra is None:
            from permutect_amd.enums import Label, Variation
            from permutect_amd.training.downsampler import (COUNT_BIN_SKIP, MAX_ALT_COUNT, MAX_REF_COUNT, NUM_ALT_COUNT_BINS,
                                                            NUM_REF_COUNT_BINS)
            shape = (self._num_sources, len(Label), len(Variation), NUM_REF_COUNT_BINS, NUM_ALT_COUNT_BINS)
            total = np.zeros(int(np.prod(shape)), dtype=np.int64)
            cols = [Data.SOURCE.idx, Data.LABEL.idx, Data.VARIANT_TYPE.idx, Data.REF_COUNT.idx, Data.ALT_COUNT.idx]
            # five of the ~58 integer columns, a slab of rows at a time: a real dataset is 10^7 - 10^8 variants of a memory MAP,
            # and every data-parallel rank fits the downsampler from these totals at start-up
            step = 1 << 20
            for lo in range(0, self._size, step):
                part = np.asarray(self._ints[lo:min(lo + step, self._size)][:, cols]).astype(np.int64)
                s, lab, v = part[:, 0], part[:, 1], part[:, 2]
                r = np.minimum(part[:, 3], MAX_REF_COUNT) // COUNT_BIN_SKIP
                a = (np.minimum(part[:, 4], MAX_ALT_COUNT) - 1) // COUNT_BIN_SKIP
                total += np.bincount(np.ravel_multi_index((s, lab, v, r, a), shape), minlength=total.size)
            self._totals_slvra = torch.from_numpy(total.astype(np.float32)).view(shape)
        return self._totals_slvra

    def num_read_features(self) -> int:
        nb = NUMBER_OF_BYTES_IN_PACKED_READ
        if self._reads is None:  # a dataset without reads (the posterior hand-off)
            return 0
        return 8 * nb + (self._reads.shape[-1] - nb) if self._reads is not None and self._reads.dtype == np.uint8 else self._reads.shape[-1]

    def num_info_features(self) -> int:
        return self._floats.shape[-1] - INFO_START_IDX

    def haplotypes_length(self) -> int:
        return self._ints.shape[-1] - HAPLOTYPES_START_IDX

    def num_sources(self) -> int:
        return self._num_sources

    def validate_sources(self) -> int:
        """Reference reads_dataset.py:212-221: every source index below the largest one must have data."""
        n = self.num_sources()
        if n == 1:
            print("Data come from a single source")
        else:
            totals_s = self.totals_slvra.sum(dim=(1, 2, 3, 4))
            for source in range(n):
                assert totals_s[source].item() >= 1, f"No data for source {source}."
            print(f"Data come from multiple sources, with counts {totals_s.tolist()}.")
        return n

    def __len__(self) -> int:
        return self._size

    # ---- sampling scheme ---------------------------------------------------------------------------------------------------
    def _chunk_ranges(self, chunk_variants: Optional[int], lo: int = 0, hi: Optional[int] = None):
        hi = self._size if hi is None else hi
        n = hi - lo
        nchunks = 1 if not chunk_variants else max(1, -(-n // chunk_variants))
        per = n // nchunks
        return [(lo + c * per, lo + (c + 1) * per if c < nchunks - 1 else hi) for c in range(nchunks)]

    def __iter__(self) -> Iterator[Datum]:
        """Datum by Datum in the reference's order of randomness (one chunk here: the whole dataset)."""
        order = np.random.permutation(self._size)
        for idx in order:
            yield Datum(self._ints[idx], self._floats[idx], self._reads[self._starts[idx]:self._starts[idx + 1]], compressed=True)

    def host_batch(self, ids: np.ndarray) -> Batch:
        """`Batch(List[Datum])` of the given dataset indices without the Datums (reference data/batch.py:41-62)."""
        ids = np.asarray(ids, dtype=np.int64)
        ints = self._ints[ids]
        floats = self._floats[ids]
        nref = ints[:, Data.REF_COUNT.idx].astype(np.int64)
        nalt = ints[:, Data.ALT_COUNT.idx].astype(np.int64)
        rows = np.concatenate([_gather_rows(self._starts, ids, np.zeros_like(nref), nref),
                               _gather_rows(self._starts, ids, nref, nalt)])
        return Batch.from_arrays(ints, floats, self._reads[rows])

    def make_data_loader(self, batch_size: int, pin_memory: bool = False, num_workers: int = 0,
                         chunk_variants: Optional[int] = None, rng: Optional[np.random.Generator] = None):
        rng = np.random.default_rng() if rng is None else rng
        dataset = self

        class _Loader:
            def __iter__(self_inner):
                ranges = dataset._chunk_ranges(chunk_variants)
                for c in rng.permutation(len(ranges)):
                    lo, hi = ranges[c]
                    order = lo + rng.permutation(hi - lo)
                    for s in range(0, len(order), batch_size):
                        b = dataset.host_batch(order[s:s + batch_size])
                        yield b.pin_memory() if pin_memory else b

            def __len__(self_inner):
                return sum(-(-(hi - lo) // batch_size) for lo, hi in dataset._chunk_ranges(chunk_variants))

        return _Loader()

    def device_loader(self, batch_size: int, device: torch.device, chunk_variants: Optional[int] = None,
                      rng: Optional[np.random.Generator] = None, shuffle: bool = True, rank: int = 0, world_size: int = 1):
        """Batches composed on the device from chunks resident in HBM.  With world_size > 1 each rank iterates its own
        contiguous shard of the dataset (reference :141-142 partitions its workers the same way)."""
        return DeviceChunkLoader(self, batch_size, device, chunk_variants, rng, shuffle, rank, world_size)


_STAGE_THREADS = 6
_PREFETCH = 3  # chunks being loaded while one is consumed: one loader thread (~15 ms of copies, packing and planning per
               # 114 MB chunk) cannot keep up with the filter forward (~6 ms per chunk of 262 144 variants)


class _UploadQueue:
    """Every host-to-device copy of the loader goes through ONE stream, one copy at a time.  Beside a busy GPU a single stream moves
    ~50 GB/s from page-locked memory, but copies in flight on several streams at once share the link at 12 - 23 GB/s
    (scripts/h2d_rate.py) -- and three prefetch threads, each with its own stream and seven copies per chunk, made exactly that
    traffic: the chunk uploads of a filter pass ran at 26 GB/s and set its pace.  `copy` enqueues on the shared stream and makes
    the CALLER's current stream wait for the copy (and the allocator know that stream)."""

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._streams = {}

    def copy(self, host: torch.Tensor, device: torch.device) -> torch.Tensor:
        if device.type != "cuda":
            return host.to(device)
        cur = torch.cuda.current_stream(device)
        with self._lock:
            up = self._streams.get(device)
            if up is None:
                up = self._streams[device] = torch.cuda.Stream(device)
            with torch.cuda.stream(up):
                out = host.to(device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(up)
        cur.wait_event(ev)
        out.record_stream(cur)
        return out


_UPLOADS = _UploadQueue()
_ONE_UPLOAD_STREAM = os.environ.get("PMT_LOADER_ONE_UPLOAD_STREAM", "1") != "0"


def _h2d(host: torch.Tensor, device: torch.device) -> torch.Tensor:
    if _ONE_UPLOAD_STREAM:
        return _UPLOADS.copy(host, device)
    return host.to(device, non_blocking=device.type == "cuda")


class PinnedStage:
    """Pinned staging buffers that live as long as their loader (pinning ~100 MB costs 10 - 70 ms, a chunk upload ~2 ms).
    One buffer per array name, grown with 12 % slack when a chunk needs more.  The user must have finished (synchronised)
    the copies out of a buffer before asking for it again."""

    def __init__(self):
        self._bufs = {}

    def get(self, name: str, shape, dtype: torch.dtype) -> torch.Tensor:
        nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        buf = self._bufs.get(name)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes * 1.125), 4096), dtype=torch.uint8, pin_memory=True)
            self._bufs[name] = buf
        return buf[:nbytes].view(dtype).view(shape)


class _StagePool:
    """Sets of pinned staging buffers that outlive a loader: a training run makes a new loader every epoch, and pinning ~115 MB
    per prefetch slot again each time cost 3 x 70 ms per epoch -- more than an epoch of 16 steps takes.  A loader borrows a set
    for its lifetime (two loaders alive at once get different sets) and hands it back when its iteration ends."""

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._free = []

    def acquire(self):
        with self._lock:
            return self._free.pop() if self._free else [PinnedStage() for _ in range(_PREFETCH)]

    def release(self, stages):
        with self._lock:
            if len(self._free) < 2:
                self._free.append(stages)


_STAGES = _StagePool()


class DeviceChunk:
    """A contiguous range of the dataset in HBM, exactly as it lies on disk."""

    def host_counts(self):
        if self.ref_host is None:
            ints = self._dataset._ints
            self.ref_host = np.asarray(ints[self.lo:self.hi, Data.REF_COUNT.idx]).astype(np.int32)
            self.alt_host = np.asarray(ints[self.lo:self.hi, Data.ALT_COUNT.idx]).astype(np.int32)
        return self.ref_host, self.alt_host

    def __init__(self, dataset: ReadsDataset, lo: int, hi: int, device: torch.device, stage: Optional[PinnedStage] = None):
        self.lo, self.hi = lo, hi
        self._dataset = dataset
        r0, r1 = int(dataset._starts[lo]), int(dataset._starts[hi])
        cuda = device.type == "cuda"
        names = iter(("ints", "floats", "reads", "row_start"))

        def upload(arr: np.ndarray) -> torch.Tensor:
            # disk (memory map) -> pinned staging buffer -> HBM; an array already in host memory is staged the same way.
            # The staging copy runs in the library on a few threads and outside the GIL (pmt_host_copy): one thread moves
            # ~10 GB/s, well under what the PCIe link takes, and Python worker threads would queue for the GIL behind
            # the training loop (5 ms per hand-off).
            dtype = torch.from_numpy(np.empty(0, arr.dtype)).dtype
            name = next(names)
            host = stage.get(name, arr.shape, dtype) if (stage is not None and cuda) else torch.empty(arr.shape, dtype=dtype, pin_memory=cuda)
            if arr.flags["C_CONTIGUOUS"] and arr.nbytes >= (1 << 22):
                L.check(L.load().pmt_host_copy(host.data_ptr(), arr.ctypes.data, arr.nbytes, _STAGE_THREADS), "pmt_host_copy")
            else:
                np.copyto(host.numpy(), arr)
            return _h2d(host, device)

        if dataset._pinned is not None and cuda:  # page-locked dataset: DMA straight from it
            ints_t, floats_t, reads_t, starts_t = dataset._pinned
            # (the read rows on a second stream -- two copy engines on one chunk -- was measured: no change, 1.00 - 1.07 ms per filter
            #  batch either way; the copies are not what a second engine would speed up)
            self.ints = _h2d(ints_t[lo:hi], device)
            self.floats = _h2d(floats_t[lo:hi], device)
            self.reads = _h2d(reads_t[r0:r1], device)
            self.row_start = _h2d(starts_t[lo:hi], device) - r0
        else:
            self.ints = upload(dataset._ints[lo:hi])                         # int16 [n, 16 + H]
            self.floats = upload(dataset._floats[lo:hi])                     # float16 [n, 6 + I]
            self.reads = upload(dataset._reads[r0:r1])                       # uint8 [R, 7 + nf]
            self.row_start = upload(dataset._starts[lo:hi] - r0)             # int64 [n]
        self.ref_host = self.alt_host = None  # the planner's counts: filled by the loader (pmt_prepare_chunk) or on demand
        self.nbytes = self.ints.numel() * 2 + self.floats.numel() * 2 + self.reads.numel() + self.row_start.numel() * 8


class ChunkBatch(Batch):
    """A batch whose variants are rows of a `DeviceChunk`: per-variant tensors gathered on the device, reads referenced
    through a gather index (no read row moves)."""

    def __init__(self, chunk: DeviceChunk, ids_host: np.ndarray, ids_dev: Optional[torch.Tensor] = None,
                 plan: Optional[GroupPlan] = None, composed=None):
        """`ids_dev` / `plan`: the ids already on the device and the batch's group plan with its device arrays, when the
        loader prepared them with the chunk (no host-to-device copy is left in the per-batch path then: a copy from
        pageable memory makes the host wait for the previous batch's kernels, so it could not run ahead of the GPU).
        `composed`: (int_tensor, float_tensor, row_start, ref_offsets, alt_offsets, read_index) already made by
        `compose_on_device` -- the loader's prefetch thread does that for every batch of a chunk right behind the chunk's upload,
        so that handing a batch to the consumer costs no launch and no torch call at all."""
        dev = chunk.ints.device
        ids = ids_dev if ids_dev is not None else torch.from_numpy(np.ascontiguousarray(ids_host, dtype=np.int64)).to(dev)
        if composed is not None:
            self.int_tensor, self.float_tensor = composed[0], composed[1]
        else:
            self.int_tensor = chunk.ints.index_select(0, ids).to(torch.long)
            self.float_tensor = chunk.floats.index_select(0, ids).to(torch.float)
        self.packed_reads = chunk.reads
        self.reads_re = None
        self._num_read_features = 8 * NUMBER_OF_BYTES_IN_PACKED_READ + chunk.reads.shape[1] - NUMBER_OF_BYTES_IN_PACKED_READ
        self._size = len(ids_host)
        # position of every variant of the batch in the dataset the loader iterates (its restricted folds): the loader orders
        # the variants INSIDE a batch for the group packer even when it does not shuffle, so a consumer that needs the
        # dataset's order (the posterior hand-off) scatters by this index
        self.dataset_index = chunk.lo + np.asarray(ids_host, dtype=np.int64)
        self.chunk_range = (chunk.lo, chunk.hi)  # the chunk's place in the dataset, and this batch's variants inside the chunk
        self.chunk_ids = ids                     # (on the device: consumers that write per-variant results in dataset order)
        self._chunk, self._ids_host = chunk, ids_host  # (the planner's counts of this batch: gathered only if somebody asks)
        self._host_counts = None
        self._total_reads = getattr(plan, "total_reads", None)
        self._plan = plan
        if composed is not None:
            self._row_start, self._offsets, self._read_index = composed[2], (composed[3], composed[4]), composed[5]
        else:
            self._offsets = None
            self._row_start = chunk.row_start.index_select(0, ids)
            self._read_index = None

    @staticmethod
    def compose_on_device(chunk: DeviceChunk, ids_dev: torch.Tensor, total_reads: int, offsets=None):
        """The batch's device tensors in ONE library call (pmt_compose_batch: gather + convert, count scans, read index) on the
        current stream; no host synchronisation, and the GIL is released for the duration of the call.  With the scans of the
        counts already on the device (`offsets`, from pmt_prepare_chunk) the call is ONE launch (pmt_compose_batch_planned)."""
        dev, n = chunk.ints.device, ids_dev.numel()
        ints = torch.empty(n, chunk.ints.shape[1], dtype=torch.long, device=dev)
        floats = torch.empty(n, chunk.floats.shape[1], dtype=torch.float32, device=dev)
        row_start = torch.empty(n, dtype=torch.long, device=dev)
        if offsets is not None:
            ref_off, alt_off = offsets
            index = torch.empty(total_reads, dtype=torch.int64, device=dev)
            L.check(L.load().pmt_compose_batch_planned(chunk.ints.data_ptr(), chunk.ints.shape[1], chunk.floats.data_ptr(), chunk.floats.shape[1],
                                                       chunk.row_start.data_ptr(), ids_dev.data_ptr(), n, ref_off.data_ptr(), alt_off.data_ptr(),
                                                       ints.data_ptr(), floats.data_ptr(), row_start.data_ptr(), index.data_ptr(),
                                                       L.raw_stream(dev)), "pmt_compose_batch_planned")
            return ints, floats, row_start, ref_off, alt_off, index
        ref_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        alt_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        index = torch.empty(total_reads, dtype=torch.int64, device=dev)
        L.check(L.load().pmt_compose_batch(chunk.ints.data_ptr(), chunk.ints.shape[1], chunk.floats.data_ptr(), chunk.floats.shape[1],
                                           chunk.row_start.data_ptr(), ids_dev.data_ptr(), n, Data.REF_COUNT.idx, Data.ALT_COUNT.idx,
                                           ints.data_ptr(), floats.data_ptr(), row_start.data_ptr(), ref_off.data_ptr(), alt_off.data_ptr(),
                                           index.data_ptr(), L.raw_stream(dev)), "pmt_compose_batch")
        return ints, floats, row_start, ref_off, alt_off, index

    def read_index(self) -> torch.Tensor:
        if self._read_index is None:
            dev = self.int_tensor.device
            lib = L.load()
            b = self._size
            stream = torch.cuda.current_stream().cuda_stream
            if self._offsets is None:
                ref_off = torch.empty(b + 1, dtype=torch.int32, device=dev)
                alt_off = torch.empty(b + 1, dtype=torch.int32, device=dev)
                ref_c, alt_c, elem, stride = self.device_counts()
                L.check(lib.pmt_scan_counts(ref_c.data_ptr(), alt_c.data_ptr(), elem, stride, b, ref_off.data_ptr(),
                                            alt_off.data_ptr(), stream), "pmt_scan_counts")
                self._offsets = (ref_off, alt_off)
            ref_off, alt_off = self._offsets
            if self._total_reads is None:
                rc, ac = self.host_counts()
                self._total_reads = int(rc.sum()) + int(ac.sum())
            total = self._total_reads
            index = torch.empty(total, dtype=torch.int64, device=dev)
            L.check(lib.pmt_build_read_index(self._row_start.data_ptr(), ref_off.data_ptr(), alt_off.data_ptr(), b,
                                             index.data_ptr(), stream), "pmt_build_read_index")
            self._read_index = index
        return self._read_index

    def host_counts(self):
        if self._host_counts is None:
            rc, ac = self._chunk.host_counts()
            self._host_counts = (rc[self._ids_host], ac[self._ids_host])
        return self._host_counts

    def get_reads_re(self) -> torch.Tensor:
        from permutect_amd.data.batch import decode_packed_reads
        rows = self.packed_reads[self.read_index()]
        return torch.from_numpy(decode_packed_reads(rows.cpu().numpy())).to(rows.device)

    def copy_to(self, device, dtype=torch.float32) -> "ChunkBatch":
        assert torch.device(device) == self.int_tensor.device, "a ChunkBatch lives on its chunk's device"
        return self

    def read_rows(self):
        return self.packed_reads, L.READS_PACKED_U8, self.packed_reads.shape[1], self.read_index()


class DeviceChunkLoader:
    def __init__(self, dataset: ReadsDataset, batch_size: int, device: torch.device, chunk_variants: Optional[int],
                 rng: Optional[np.random.Generator], shuffle: bool, rank: int, world_size: int):
        self.dataset, self.batch_size, self.device = dataset, batch_size, torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:  # (`torch.device("cuda")`: the prefetch threads need the card spelled out)
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.rng = np.random.default_rng() if rng is None else rng
        self.shuffle = shuffle
        n = len(dataset)
        per = n // world_size  # contiguous shard per rank; the last rank takes the remainder
        self.lo = rank * per
        self.hi = (rank + 1) * per if rank < world_size - 1 else n
        self.ranges = dataset._chunk_ranges(chunk_variants, self.lo, self.hi)
        # a short first chunk: the consumer starts after ONE batch's worth of upload and preparation instead of a whole chunk's
        # (~10 ms of a 5 M-candidate filter pass that takes 90), while the full-size chunks behind it are already on their way
        # (round 4: and that first piece is loaded ALONE before the prefetch threads start on the others (__iter__): starting together
        #  they put the first batch's seven small copies behind the other chunks' large ones on the one upload stream, and the first
        #  batch of a pass left the loader after 6.5 ms instead of ~1.5)
        first_lo, first_hi = self.ranges[0]
        if len(self.ranges) > 1 and first_hi - first_lo >= 4 * batch_size:  # (measured: loading the short piece ALONE first made the SECOND batch late: all prefetch threads start at once)
            self.ranges = [(first_lo, first_lo + batch_size), (first_lo + batch_size, first_hi)] + self.ranges[1:]
        self.bytes_uploaded = 0
        self._stages = None  # borrowed from _STAGES while iterating: one PinnedStage per chunk in flight
        self._slot_events = [None] * _PREFETCH  # per staging slot: the event behind the last chunk enqueued out of it
        self._compose_on_consumer = os.environ.get("PMT_LOADER_COMPOSE", "consumer") != "prefetch"

    def __len__(self) -> int:
        return sum(-(-(hi - lo) // self.batch_size) for lo, hi in self.ranges)

    def _prepare_fast(self, chunk: DeviceChunk, seed: int, stage: PinnedStage):
        """`_prepare` in one GIL-free library call (pmt_prepare_chunk): counts, consumption order and every batch's group plan
        written straight into the pinned buffer that is then uploaded with one copy.  None when the chunk holds a read set
        beyond one workgroup (the split planner takes such a chunk batch by batch: `_prepare`)."""
        dev, bs, ds = self.device, self.batch_size, self.dataset
        ints = ds._ints
        if not (isinstance(ints, np.ndarray) and ints.dtype == np.int16 and ints.strides[1] == 2 and ints.strides[0] % 2 == 0):
            return None
        n = chunk.hi - chunk.lo
        nb = -(-n // bs)
        cap = 2 * (n + nb)
        total = 2 * n + cap  # ids (int64 = 2 ints each) first: stays 8-byte aligned
        host = stage.get("plans", (total,), torch.int32) if dev.type == "cuda" else torch.empty(total, dtype=torch.int32)
        flat = host.numpy()
        ref_host, alt_host = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        info = np.zeros((nb, 4), dtype=np.int32)
        # per batch the exclusive scans of its counts (ref | alt, bs + 1 each): composing a batch on the device is then ONE launch
        offs = (stage.get("offsets", (nb, 2, bs + 1), torch.int32) if dev.type == "cuda" else torch.empty((nb, 2, bs + 1), dtype=torch.int32))
        rc = L.load().pmt_prepare_chunk(ints[chunk.lo:].ctypes.data, ints.strides[0] // 2, Data.REF_COUNT.idx, Data.ALT_COUNT.idx, n,
                                        1 if self.shuffle else 0, seed & 0xFFFFFFFFFFFFFFFF, bs, 64, 8, ref_host.ctypes.data, alt_host.ctypes.data,
                                        flat.ctypes.data, flat[2 * n:].ctypes.data, cap, info.ctypes.data, offs.data_ptr())
        if rc == L.E_CAPACITY:
            return None
        L.check(rc, "pmt_prepare_chunk")
        chunk.ref_host, chunk.alt_host = ref_host, alt_host
        used = 2 * n + int(info[nb - 1, 0]) + 2 * (int(info[nb - 1, 1]) + 1) if nb else 2 * n
        flat_dev = _h2d(host[:used], dev)
        offs_dev = _h2d(offs, dev)
        ids_host = flat[: 2 * n].view(np.int64).copy()  # (the pinned buffer is reused for the next chunk of this slot)
        ids_dev = flat_dev[: 2 * n].view(torch.int64)
        out = []
        for k in range(nb):
            at, g = 2 * n + int(info[k, 0]), int(info[k, 1])
            gs_h, gt_h = flat[at:at + g + 1].copy(), flat[at + g + 1:at + 2 * g + 2].copy()
            plan = GroupPlan.from_prepared(gs_h, gt_h, {str(dev): (flat_dev[at:at + g + 1], flat_dev[at + g + 1:at + 2 * g + 2], None)})
            plan.total_reads = int(info[k, 3])
            m = min(bs, n - k * bs)
            plan.offsets_dev = (offs_dev[k, 0, :m + 1], offs_dev[k, 1, :m + 1])
            out.append((ids_host[k * bs:k * bs + m], ids_dev[k * bs:k * bs + m], plan))
        chunk.plans_dev = flat_dev
        chunk.offsets_dev = offs_dev
        return out

    def _prepare(self, chunk: DeviceChunk, ids: np.ndarray, stage: PinnedStage):
        """Everything the chunk's batches need from the host, uploaded ONCE with the chunk: the shuffled variant ids and
        every batch's group plan (one pinned buffer, one copy); returns per batch (ids_host, ids_dev, plan)."""
        dev, bs = self.device, self.batch_size
        chunk.host_counts()
        # inside a batch the variants go in the order that fills the workgroups best (the batch is a random draw anyway):
        # one GIL-free call for all batches of the chunk
        rc, ac = np.ascontiguousarray(chunk.ref_host[ids]), np.ascontiguousarray(chunk.alt_host[ids])
        order = np.empty(len(ids), dtype=np.int32)
        L.check(L.load().pmt_pack_order_batches(rc.ctypes.data, ac.ctypes.data, len(ids), bs, 64, 4, order.ctypes.data), "pmt_pack_order_batches")
        ids = ids[order]
        slices = [ids[s:s + bs] for s in range(0, len(ids), bs)]
        plans = [GroupPlan(chunk.ref_host[sl], chunk.alt_host[sl], allow_split=True) for sl in slices]
        parts = []
        for p in plans:
            parts += [p.group_start, p.group_tile_base] + ([] if p.span is None else [p.span.ravel()])
        flat = np.concatenate([ids.astype(np.int64).view(np.int32)] + parts)  # ids first: stays 8-byte aligned
        host = stage.get("plans", flat.shape, torch.int32) if dev.type == "cuda" else torch.empty(flat.shape, dtype=torch.int32)
        host.numpy()[...] = flat
        flat_dev = _h2d(host, dev)
        ids_dev = flat_dev[: 2 * len(ids)].view(torch.int64)
        at = 2 * len(ids)
        out = []
        for k, (sl, p) in enumerate(zip(slices, plans)):
            views = []
            for arr in (p.group_start, p.group_tile_base):
                views.append(flat_dev[at:at + arr.size])
                at += arr.size
            span = None
            if p.span is not None:
                span = flat_dev[at:at + p.span.size].view(p.span.shape)
                at += p.span.size
            p._dev[str(dev)] = (views[0], views[1], span)
            out.append((sl, ids_dev[k * bs:k * bs + len(sl)], p))
        chunk.plans_dev = flat_dev  # (the ids and plans of its batches are views of this one allocation)
        return out

    def _load(self, c: int, seed: int, slot: int):
        """Runs on a prefetch thread: stage chunk c, ENQUEUE its upload, its batches' ids / plans and their composition on a side
        stream, and return the event behind them -- the consumer's stream waits for that event on the device, its thread never
        does (it used to block until the chunk was resident and then needed 0.1 - 0.35 ms to wake up and launch while the GPU idled
        at every chunk boundary).  `slot` names the pinned staging buffers: before they are written again, this thread waits for
        the event of the chunk that used them last."""
        stage = self._stages[slot]
        last = self._slot_events[slot]
        if last is not None:
            last.synchronize()  # (the copies out of this slot's buffers have left)
        lo, hi = self.ranges[c]
        n = hi - lo

        def prepare(chunk):
            fast = self._prepare_fast(chunk, seed, stage)
            if fast is not None:
                return fast
            ids = np.random.default_rng(seed).permutation(n) if self.shuffle else np.arange(n)
            return self._prepare(chunk, ids, stage)
        if self.device.type != "cuda":
            chunk = DeviceChunk(self.dataset, lo, hi, self.device)
            return chunk, prepare(chunk), None
        torch.cuda.set_device(self.device)
        # (a HIGH-priority stream for the chunk's copies and composition kernels was measured SLOWER: 1.09 - 1.12 against 1.01 - 1.05 ms
        #  per filter batch -- its kernels then displace the consumer's read-set kernels instead of filling their tails)
        side = torch.cuda.Stream(self.device)
        import time
        t0 = time.perf_counter()
        # (every chunk's upload on ONE stream shared by the prefetch threads -- scripts/h2d_rate.py shows copies from several streams
        #  sharing the link badly beside a busy GPU -- was measured: no change, 1.01 - 1.08 ms per filter batch either way)
        with torch.cuda.stream(side):
            chunk = DeviceChunk(self.dataset, lo, hi, self.device, stage)
        with torch.cuda.stream(side):
            t1 = time.perf_counter()
            batches = prepare(chunk)
            # ... and every batch of the chunk composed right here, on this thread's stream, behind the upload
            composed = []
            for ids_host, ids_dev, plan in batches:
                if self._compose_on_consumer and getattr(plan, "offsets_dev", None) is not None and getattr(plan, "total_reads", None) is not None:
                    composed.append(None)  # one launch on the consumer's own stream, in front of the batch's kernels (__iter__)
                    continue
                total = getattr(plan, "total_reads", None)
                if total is None:
                    rc, ac = chunk.host_counts()
                    total = int(rc[ids_host].sum()) + int(ac[ids_host].sum())
                composed.append(ChunkBatch.compose_on_device(chunk, ids_dev, total, getattr(plan, "offsets_dev", None)))
            batches = [b + (c,) for b, c in zip(batches, composed)]
            done = torch.cuda.Event()
            done.record(side)
        self._slot_events[slot] = done
        t2 = time.perf_counter()
        if os.environ.get("PMT_LOADER_TIMING"):
            print(f"[loader] chunk {c}: stage+enqueue {1e3 * (t1 - t0):.1f} ms, prepare + compose enqueued {1e3 * (t2 - t1):.1f} ms", flush=True)
        return chunk, batches, done

    def __iter__(self) -> Iterator[ChunkBatch]:
        """One chunk is trained on while the next ones are read, staged and uploaded by background threads."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        import time as _time
        _t_iter = _time.perf_counter()
        _timing = bool(os.environ.get("PMT_LOADER_TIMING"))
        order_c = self.rng.permutation(len(self.ranges)) if self.shuffle else np.arange(len(self.ranges))
        seeds = self.rng.integers(0, 2 ** 63 - 1, size=len(order_c))  # one stream per chunk: the prefetch thread shuffles
        self._stages = _STAGES.acquire()
        # Compose one batch AHEAD on a side stream: the dozen small gather / scan launches of batch i + 1 (~0.1 ms of device time)
        # then run beside the read-set kernels of batch i instead of in front of its own.  And while batches are being drawn, the
        # interpreter hands the GIL over every 0.5 ms instead of every 5: the prefetch threads run short stretches of Python between
        # their GIL-free library calls, and the consumer's ~40 launches per step must not queue behind them for milliseconds.
        import sys
        cuda = self.device.type == "cuda"
        compose = torch.cuda.Stream(self.device) if cuda else None
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, float(os.environ.get("PMT_LOADER_SWITCH_INTERVAL", "5e-4"))))

        def composed(chunk, ids_host, ids_dev, plan, ready_made=None):
            if not cuda:
                return ChunkBatch(chunk, ids_host, ids_dev, plan), None
            if ready_made is not None:  # composed by the prefetch thread behind the chunk's upload (which _load waited for)
                cur = torch.cuda.current_stream(self.device)
                for t in ready_made:
                    t.record_stream(cur)
                return ChunkBatch(chunk, ids_host, ids_dev, plan, composed=ready_made), None
            if self._compose_on_consumer and getattr(plan, "offsets_dev", None) is not None and getattr(plan, "total_reads", None) is not None:
                # ONE launch (pmt_compose_batch_planned) on the consumer's stream: 0.04 ms in front of the batch's own kernels.  Beside
                # them, on another stream, the same launch cost the read-set forward 0.07 ms: its 3 400 workgroups run in 6.7 rounds
                # of 512, and workgroups of another kernel that take slots in between push it into one more partial round
                made = ChunkBatch.compose_on_device(chunk, ids_dev, plan.total_reads, plan.offsets_dev)
                return ChunkBatch(chunk, ids_host, ids_dev, plan, composed=made), None
            cur = torch.cuda.current_stream(self.device)
            compose.wait_stream(cur)  # (nothing of the consumer's stream may be overtaken by a reuse of freed memory)
            with torch.cuda.stream(compose):
                cb = ChunkBatch(chunk, ids_host, ids_dev, plan)
                cb.read_index()
                ev = torch.cuda.Event()
                ev.record(compose)
            for t in (cb.int_tensor, cb.float_tensor, cb._row_start, cb._read_index, *cb._offsets):
                t.record_stream(cur)  # allocated on the side stream, consumed on the caller's
            return cb, ev

        def ready(item):
            cb, ev = item
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
            return cb
        try:
            with ThreadPoolExecutor(max_workers=_PREFETCH) as pool:
                pending = deque()

                def submit(i):
                    if i < len(order_c):
                        pending.append(pool.submit(self._load, int(order_c[i]), int(seeds[i]), i % _PREFETCH))
                # _PREFETCH loads in flight
                submitted = 0
                for i in range(max(1, min(_PREFETCH, len(order_c)))):
                    submit(i)
                    submitted += 1
                ahead = None
                if _timing:
                    print(f"[loader] iteration set up and first loads submitted after {1e3 * (_time.perf_counter() - _t_iter):.2f} ms", flush=True)
                for i in range(len(order_c)):
                    chunk, batches, done = pending.popleft().result()
                    if _timing and i < 3:
                        print(f"[loader] chunk {i} ({len(batches)} batches) in hand after {1e3 * (_time.perf_counter() - _t_iter):.2f} ms", flush=True)
                    want = i + 1 + _PREFETCH
                    while submitted < min(want, len(order_c)):
                        submit(submitted)
                        submitted += 1
                    self.bytes_uploaded += chunk.nbytes
                    if cuda:
                        torch.cuda.current_stream(self.device).wait_event(done)  # the chunk and its composed batches, on the device
                        compose.wait_event(done)
                        # the chunk was allocated on the prefetch thread's side stream and is consumed on THIS stream: tell the
                        # allocator, or a chunk dropped while its last batch's kernels are still queued could be handed out again
                        cur = torch.cuda.current_stream(self.device)
                        for t in (chunk.ints, chunk.floats, chunk.reads, chunk.row_start, getattr(chunk, "plans_dev", None),
                                  getattr(chunk, "offsets_dev", None)):
                            if t is not None:
                                t.record_stream(cur)
                                t.record_stream(compose)
                    for item in batches:
                        nxt = composed(chunk, *item)
                        if nxt[1] is None and ahead is None:
                            # composed on the consumer's own stream: ready as it is.  (Held back as `ahead` like a side-stream batch,
                            # the single batch of the short first chunk left the loader only once the SECOND chunk had been loaded:
                            # the first batch of a pass came after 6.5 ms, the short chunk bought nothing)
                            yield nxt[0]
                            continue
                        if ahead is not None:
                            yield ready(ahead)
                        ahead = nxt
                if ahead is not None:
                    yield ready(ahead)
        finally:  # (the executor has joined its threads; wait for the copies they enqueued out of the staging buffers)
            for ev in self._slot_events:
                if ev is not None:
                    ev.synchronize()
            self._slot_events = [None] * _PREFETCH
            sys.setswitchinterval(old_interval)
            stages, self._stages = self._stages, None
            _STAGES.release(stages)
