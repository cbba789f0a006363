"""The artifact model's hand-off to the posterior model in `filter_variants` (reference tools/filter_variants.py:292-320,
:350-357): every candidate becomes a Datum WITHOUT reads whose info array is its embedding and whose
CACHED_ARTIFACT_LOGIT is the model's logit.

The reference does this one variant at a time in Python (`logits_b.tolist()`, `features_be.cpu()`, a Datum per variant,
amortised-growth memory maps): at 5 M candidates that loop dwarfs the GPU forward.  Here a batch's rows are transformed
as arrays on the device and appended to preallocated host arrays with one copy per batch."""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Optional

import numpy as np
import torch

from permutect_amd.data.datum import Data, INFO_START_IDX
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.data.reads_dataset import ReadsDataset


def posterior_rows(int_tensor: torch.Tensor, float_tensor: torch.Tensor, logits_b: torch.Tensor, features_be: torch.Tensor):
    """(int16 [B, 16 + H], float32 [B, 6 + E]) exactly as the reference's per-Datum code leaves them:
    counts zeroed; the logit stored through the float16 scalar array (`Datum.set`, reference data/datum.py:199-211), then
    `set_info_1d(embedding)` replaces the info columns by the float32 embedding, which promotes the row to float32."""
    ints = int_tensor.to(torch.int16).clone()
    ints[:, Data.REF_COUNT.idx] = 0
    ints[:, Data.ALT_COUNT.idx] = 0
    scalars = float_tensor[:, :INFO_START_IDX].to(torch.float16)
    scalars[:, Data.CACHED_ARTIFACT_LOGIT.idx] = logits_b.to(torch.float16)
    floats = torch.cat((scalars.to(torch.float32), features_be.to(torch.float32)), dim=1)
    return ints, floats


def device_posterior_rows(float_tensor: torch.Tensor, logits_b: torch.Tensor, features_be: torch.Tensor, dest_ids: Optional[torch.Tensor],
                          block: torch.Tensor):
    """The float rows of `posterior_rows` for a batch on the device, written to rows `dest_ids` (None: in order) of `block`, in ONE
    launch (pmt_posterior_rows) -- was five torch launches per batch in a loop whose host time exceeded the forward it feeds.  On
    the CPU (no library call without a GPU) the same rows by torch."""
    if float_tensor.device.type != "cuda":
        scalars = float_tensor[:, :INFO_START_IDX].to(torch.float16)
        scalars[:, Data.CACHED_ARTIFACT_LOGIT.idx] = logits_b.to(torch.float16)
        rows = torch.cat((scalars.to(torch.float32), features_be.to(torch.float32)), dim=1)
        if dest_ids is None:
            block[: rows.shape[0]] = rows
        else:
            block.index_copy_(0, dest_ids, rows)
        return
    from permutect_amd.engine import lib as L
    assert float_tensor.dtype == torch.float32 and float_tensor.stride(1) == 1 and block.dtype == torch.float32 and block.stride(1) == 1
    logits_b, features_be = logits_b.contiguous().float(), features_be.contiguous().float()
    assert dest_ids is None or (dest_ids.dtype == torch.int64 and dest_ids.is_contiguous())
    n, e = logits_b.shape[0], features_be.shape[1]
    assert block.shape[1] == INFO_START_IDX + e
    L.check(L.load().pmt_posterior_rows(float_tensor.data_ptr(), float_tensor.stride(0), INFO_START_IDX, Data.CACHED_ARTIFACT_LOGIT.idx,
                                        logits_b.data_ptr(), features_be.data_ptr(), e, None if dest_ids is None else dest_ids.data_ptr(), n,
                                        block.data_ptr(), block.stride(0), torch.cuda.current_stream().cuda_stream), "pmt_posterior_rows")


@torch.inference_mode()
def make_posterior_mmap(dataset: ReadsDataset, model, batch_size: int, device: Optional[torch.device] = None,
                        chunk_variants: Optional[int] = None, rank: int = 0, world_size: int = 1) -> Optional[MemoryMappedData]:
    """`MemoryMappedData.from_generator(generate_posterior_data(...))` of the reference, in dataset order.

    `rank`, `world_size` (one process per GPU, torch.distributed initialised by the caller: tools/filter_variants.py under torchrun):
    the candidate index range is cut into `world_size` contiguous shards (the reference's loader partitions its workers
    the same way, data/reads_dataset.py:141-142), every rank runs the forward over ITS shard with no collective on the data path, and
    the shards' rows are concatenated on rank 0 in dataset order (point-to-point over a host-side gloo group, straight into their
    place in the result: the rows are already in host memory).  Rank 0 returns the whole result -- bit for bit the single-process one
    (tests/test_distributed_cpu.py) -- the other ranks None.

    Nothing waits for the device inside the loop.  The float rows (six scalars with the logit stored through float16, then the
    float32 embedding: `posterior_rows`) are assembled on the device and scattered to their place inside a per-CHUNK device
    block (the loader orders the variants of a batch for the group packer, also without shuffling); a finished chunk's block
    goes to the host as ONE contiguous copy into a pinned buffer, and a helper thread moves it into the result once the copy's
    event has fired.  The integer rows never touch the device: they are the dataset's own rows with the two counts zeroed,
    copied (and cleared) in ONE pass by a second helper thread while the GPU works (pmt_host_copy_rows).  Measured on 5 x 2^20
    candidates: the device pass 95 ms, integer rows 8 ms, float-row copies 12 ms, everything home 112 ms after the start."""
    import threading
    from collections import deque
    from queue import Queue
    from permutect_amd.engine import lib as L
    device = model._device if device is None else torch.device(device)
    if device.type == "cuda":
        dataset.pin_memory_if_it_fits()  # (chunks by DMA straight from the dataset when it fits host memory; else staged copies)
    n_total = len(dataset)
    loader = dataset.device_loader(batch_size, device, chunk_variants=chunk_variants, shuffle=False, rank=rank, world_size=world_size)
    shard_lo, shard_hi = loader.lo, loader.hi  # this rank's contiguous share of the candidates
    # rank 0 holds the whole result (its shard is the first), the others their shard only; `base`: dataset index of row 0
    base, n = (0, n_total) if rank == 0 else (shard_lo, shard_hi - shard_lo)
    e = model.reducer.output_dimension()
    width = INFO_START_IDX + e
    ints_out = np.empty((n, dataset._ints.shape[-1]), dtype=np.int16)
    floats_out = np.empty((n, width), dtype=np.float32)
    timing = os.environ.get("PMT_POSTERIOR_TIMING")
    t_start = time.perf_counter()
    t_ints = [0.0]
    t_floats = [0.0]
    model.train(False)
    cuda = device.type == "cuda"
    lib = L.load()

    def host_copy(dst: np.ndarray, src_ptr: int, nbytes: int):
        if nbytes >= (1 << 22):
            L.check(lib.pmt_host_copy(dst.ctypes.data, src_ptr, nbytes, 12), "pmt_host_copy")
        else:
            C.memmove(dst.ctypes.data, src_ptr, nbytes)

    jobs: Queue = Queue()
    errors = []

    def ints_worker():  # the integer rows: the dataset's own, counts zeroed (never on the device)
        t0 = time.perf_counter()
        try:
            mine = dataset._ints[shard_lo:shard_hi]
            src = mine if mine.flags["C_CONTIGUOUS"] else np.ascontiguousarray(mine)
            assert Data.ALT_COUNT.idx == Data.REF_COUNT.idx + 1
            L.check(lib.pmt_host_copy_rows(ints_out[shard_lo - base:].ctypes.data, src.ctypes.data, shard_hi - shard_lo, ints_out.shape[1] * 2,
                                           Data.REF_COUNT.idx * 2, 4, 8), "pmt_host_copy_rows")  # one pass: the rows and the two zeroed counts
            t_ints[0] = time.perf_counter() - t0
        except Exception as exc:
            errors.append(exc)

    def worker():       # the float rows: chunk blocks as their device-to-host copies complete
        try:
            while True:
                job = jobs.get()
                if job is None:
                    return
                event, pinned, lo, hi, free = job
                if event is not None:
                    event.synchronize()
                t0 = time.perf_counter()
                host_copy(floats_out[lo - base:hi - base], pinned.data_ptr(), (hi - lo) * width * 4)
                t_floats[0] += time.perf_counter() - t0
                free.append(pinned)
        except Exception as exc:  # surfaced by the caller after join
            errors.append(exc)

    th = threading.Thread(target=worker, daemon=True)
    th.start()
    th_ints = threading.Thread(target=ints_worker, daemon=True)
    th_ints.start()
    free_pinned: deque = deque()
    block, block_range, done = None, None, 0
    d2h_stream = torch.cuda.Stream(device) if cuda else None

    def flush():
        nonlocal block
        if block is None:
            return
        lo, hi = block_range
        if cuda:
            while not free_pinned and th.is_alive() and jobs.qsize() >= 4:  # at most four chunk blocks in flight
                th.join(0.0005)
            pinned = free_pinned.popleft() if free_pinned else torch.empty(block.shape, dtype=torch.float32, pin_memory=True)
            if pinned.shape[0] < hi - lo:
                pinned = torch.empty(block.shape, dtype=torch.float32, pin_memory=True)
            # the copy home runs on a stream of its own: on the compute stream its ~0.7 ms per chunk (17 MB over PCIe) stood between
            # two batches' kernels -- a fifth of the pass.  It waits for the block's last writer and nothing waits for it but the
            # helper thread (the block's memory is kept from reuse until the copy is done: record_stream)
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(d2h_stream):
                d2h_stream.wait_event(ready)
                pinned[: hi - lo].copy_(block, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(d2h_stream)
            block.record_stream(d2h_stream)
            jobs.put((ev, pinned, lo, hi, free_pinned))
        else:
            jobs.put((None, block, lo, hi, deque()))
        block = None

    for batch in loader:
        out = model.compute_batch_output(batch)
        lo, hi = batch.chunk_range
        if block_range != (lo, hi):
            flush()
            block, block_range = torch.empty(hi - lo, width, dtype=torch.float32, device=device), (lo, hi)
        device_posterior_rows(batch.float_tensor, out.logits_b, out.features_be, batch.chunk_ids, block)
        done += batch.size()
    flush()
    t_enqueued = time.perf_counter()
    jobs.put(None)
    th.join()
    t_floats_done = time.perf_counter()
    th_ints.join()
    if timing:
        print(f"[posterior] {n} rows: loop enqueued {1e3 * (t_enqueued - t_start):.1f} ms, float rows home {1e3 * (t_floats_done - t_start):.1f} ms "
              f"(copies {1e3 * t_floats[0]:.1f}), integer rows {1e3 * t_ints[0]:.1f} ms, all {1e3 * (time.perf_counter() - t_start):.1f} ms", flush=True)
    if errors:
        raise errors[0]
    assert done == shard_hi - shard_lo
    model.engine().check_join_fault()  # (every float row is home: the device has finished; a timed-out join means wrong logits -> raise)
    if world_size > 1:  # the shards' rows, in dataset order, on rank 0 (no collective on the data path: this is the hand-over of results)
        import torch.distributed as dist
        from permutect_amd.training.distributed import host_group
        group = host_group()
        # (every rank got here, i.e. its shard went through: a rank that raised above has left the others at this all-reduce, which fails
        #  with it -- instead of rank 0 waiting for rows that will never be sent)
        done_everywhere = torch.ones(1, dtype=torch.int32)
        dist.all_reduce(done_everywhere, op=dist.ReduceOp.MIN, group=group)
        per = n_total // world_size
        if rank == 0:
            for r in range(1, world_size):
                lo, hi = r * per, ((r + 1) * per if r < world_size - 1 else n_total)
                if hi > lo:
                    dist.recv(torch.from_numpy(ints_out[lo:hi]), src=r, group=group)
                    dist.recv(torch.from_numpy(floats_out[lo:hi]), src=r, group=group)
        else:
            if n > 0:
                dist.send(torch.from_numpy(ints_out), dst=0, group=group)
                dist.send(torch.from_numpy(floats_out), dst=0, group=group)
            return None
    return MemoryMappedData(ints_out, floats_out, n, None, 0)
