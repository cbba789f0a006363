"""TEST INFRASTRUCTURE (moved out of the product package in round 4): a whole training step as ONE captured HIP graph.

It proves a property of the C ABI -- every entry point of a training step is stream-ordered, allocates nothing and reads what varies
from device memory, so the step can be captured and replayed (tests/test_graph_gpu.py) -- and it is kept for that.  As a PRODUCT path
it earned nothing: at B = 64 / 8 192 a replay (0.68 / 1.09 ms with the batch upload) does not beat the eager step (0.54 - 0.71 /
1.11 ms); a step's ~27 kernels are a dependent chain whose device time (0.46 / 0.83 ms) is the floor either way, and hipGraph's
per-node launch cost is not below the stream's.  The original description:

A whole training step as ONE captured HIP graph, for the batch sizes where the host is the bottleneck.

At the reference's default training batch (64 read sets, parameters.py:214) a step is ~0.2 ms of kernels inside ~0.6-0.9 ms
of Python, autograd and launch overhead (~45 launches).  `GraphedTrainStep` captures zero_grad -> forward -> losses ->
backward -> clip + AdamW once and replays it; what varies from batch to batch lives in DEVICE memory so that the captured
launches (fixed grids, fixed pointers) stay valid:

  * the batch itself: `StaticBatch` = fixed-capacity device buffers (per-variant tables, packed reads, group plan) that
    `load()` refills with asynchronous copies; the count scans run inside the graph;
  * the number of workgroups: the read-set kernels are launched for the plan's CAPACITY and read the batch's real group
    count from `PmtBatch.num_groups_dev` (workgroups beyond it return at once);
  * the optimizer's step number: `PMT_STEP_ON_DEVICE` (the launch increments a device counter itself).

Not covered (the eager path runs these): read sets beyond one workgroup (layered execution), batches that exceed the
capacity, a balancer between forward and losses (its weights depend on the forward's output through host logic), data
parallel reduction.  A learning-rate change re-captures.
"""
from __future__ import annotations

from typing import Optional

import torch

from permutect_amd.data.batch import Batch
from permutect_amd.data.datum import NUMBER_OF_BYTES_IN_PACKED_READ
from permutect_amd.engine import lib as L


class _StaticPlan:
    """What the engine asks of a plan, with device arrays that never move and a group count equal to the capacity."""

    layered = False
    span = None

    def __init__(self, max_groups: int, max_tiles: int, device: torch.device):
        self.num_groups = max_groups
        self.total_tiles = max_tiles
        self.group_start = torch.zeros(max_groups + 1, dtype=torch.int32, device=device)
        self.group_tile_base = torch.zeros(max_groups + 1, dtype=torch.int32, device=device)
        self.num_groups_dev = torch.zeros(1, dtype=torch.int32, device=device)

    def on(self, device):
        return self.group_start, self.group_tile_base, None


class StaticBatch(Batch):
    """Fixed-capacity device buffers in the Batch layout.  `load(batch)` copies a host (or device) batch of exactly
    `batch_size` variants and at most `max_reads` packed read rows into them."""

    def __init__(self, batch_size: int, max_reads: int, device: torch.device, int_cols: int, float_cols: int, read_bytes: int = 12,
                 max_groups: Optional[int] = None):
        self._size = batch_size
        self.device = torch.device(device)
        self.int_tensor = torch.zeros(batch_size, int_cols, dtype=torch.long, device=device)
        self.int_tensor[:, 1] = 1  # a valid (if empty) batch until the first load: one alt read per variant
        self.float_tensor = torch.zeros(batch_size, float_cols, dtype=torch.float32, device=device)
        self.max_reads = max(max_reads, batch_size)
        self.packed_reads = torch.zeros(self.max_reads, read_bytes, dtype=torch.uint8, device=device)
        self.reads_re = None
        self._num_read_features = 8 * NUMBER_OF_BYTES_IN_PACKED_READ + read_bytes - NUMBER_OF_BYTES_IN_PACKED_READ
        if max_groups is None:  # next-fit packing of wave-sized pieces: <= reads / 128 + B / 2 groups, and one per 64 sets
            max_groups = min(batch_size, self.max_reads // 128 + batch_size // 2 + batch_size // 64 + 2)
        max_tiles = min(max_groups * L.GROUP_TILES, (self.max_reads + 15) // 16 + 2 * max_groups)
        self._plan = _StaticPlan(max_groups, max_tiles, self.device)
        # (until the first load() the batch has no groups: every workgroup of a launch returns at once)
        self._offsets = None
        self._host_counts = None
        self._stage = {}

    def plan(self, allow_split: bool = False, fault=None):
        return self._plan

    def host_counts(self):
        raise L.PmtError("a StaticBatch has no host-side counts: plan on the batch that is loaded into it")

    _RING = 4  # pinned staging slots for the group plan: a slot is rewritten only after the copies out of it have run

    def _plan_slot(self, n_ints: int):
        """The next pinned staging slot (int32, capacity for any plan of this batch), free to overwrite: `load` records an event
        behind the copies it issues from a slot and waits on that event before the slot's next use -- the host may run up to
        _RING loads ahead of the device and never rewrites bytes a queued copy has yet to read (an earlier version staged every
        load in ONE buffer: a replay could then see the NEXT batch's group plan)."""
        if not self._stage:
            cap = 2 * (self._plan.num_groups + 1) + 1
            self._stage = {"bufs": [torch.empty(cap, dtype=torch.int32, pin_memory=True) for _ in range(self._RING)],
                           "events": [None] * self._RING, "at": 0}
        st = self._stage
        slot = st["at"] % self._RING
        st["at"] += 1
        if st["events"][slot] is not None:
            st["events"][slot].synchronize()
        assert n_ints <= st["bufs"][slot].numel()
        return slot, st["bufs"][slot]

    def load(self, src: Batch):
        """Refill the buffers from `src` (a Batch with packed reads, host or device, no gather index).  Asynchronous on the
        current stream; raises if `src` does not fit the capacity (the caller then steps eagerly).  The caller must not rewrite
        `src`'s own (pinned) tensors before the copies have run; the plan staging in here is safe to run ahead (see _plan_slot)."""
        if src.size() != self._size or src.packed_reads is None or getattr(src, "read_index", None) is not None:
            raise L.PmtError("StaticBatch.load wants a batch of the captured size with packed reads and no gather index")
        plan = src.plan()  # host arrays (raises for read sets beyond one workgroup)
        r = src.packed_reads.shape[0]
        if r > self.max_reads or plan.num_groups > self._plan.num_groups or plan.total_tiles > self._plan.total_tiles:
            raise L.PmtError(f"batch exceeds the captured capacity ({r} reads, {plan.num_groups} groups, {plan.total_tiles} tiles)")
        nb = True
        self.int_tensor.copy_(src.int_tensor, non_blocking=nb)
        self.float_tensor.copy_(src.float_tensor, non_blocking=nb)
        self.packed_reads[:r].copy_(src.packed_reads, non_blocking=nb)
        g = plan.num_groups
        slot, buf = self._plan_slot(2 * (g + 1) + 1)
        host = buf.numpy()
        host[: g + 1] = plan.group_start[: g + 1]
        host[g + 1: 2 * g + 2] = plan.group_tile_base[: g + 1]
        host[2 * g + 2] = g
        self._plan.group_start[: g + 1].copy_(buf[: g + 1], non_blocking=nb)
        self._plan.group_tile_base[: g + 1].copy_(buf[g + 1: 2 * g + 2], non_blocking=nb)
        self._plan.num_groups_dev.copy_(buf[2 * g + 2: 2 * g + 3], non_blocking=nb)
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._stage["events"][slot] = ev
        return self


class GraphedTrainStep:
    """`step = GraphedTrainStep(model, optimizer, static_batch)`, then per batch: `static_batch.load(b); loss = step()`.
    The returned tensors are the graph's static outputs (valid until the next replay)."""

    def __init__(self, model, optimizer, batch: StaticBatch, warmup: int = 2):
        self.model, self.opt, self.batch = model, optimizer, batch
        if model.engine().plan.desc.dropout_p > 0:
            # the step's dropout seed is a launch argument: a captured graph would replay ONE mask for ever
            raise NotImplementedError("a model with dropout_p > 0 trains eagerly (a new seed per step); GraphedTrainStep would freeze the masks")
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._lr = None
        self._warmup = warmup
        self._device_steps = -1
        self.total_loss = None
        self.logits_b = None

    def _eager(self):
        self.opt.zero_grad()
        out = self.model.compute_batch_output(self.batch)
        losses = self.model.compute_batch_losses(out, self.batch)
        losses.total_loss.backward()
        self.opt.step(step_on_device=True)
        return out, losses

    def _capture(self):
        self.model.train(True)
        eng = self.model.engine()
        if eng.grad_hook is not None:
            raise L.PmtError("GraphedTrainStep does not cover the data-parallel gradient reduction")
        self.opt._bind()
        # the warm-up steps below must leave no trace: parameters, moments and step count are put back afterwards
        saved = [t.clone() for t in (eng.space.theta, self.opt.exp_avg, self.opt.exp_avg_sq)]
        side = torch.cuda.Stream(self.batch.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # torch's recipe: warm up on a side stream so that capture sees a settled allocator
            for _ in range(self._warmup):
                self.batch._offsets = None
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for dst, src in zip((eng.space.theta, self.opt.exp_avg, self.opt.exp_avg_sq), saved):
                dst.copy_(src)
        self.opt.set_device_step(self.opt.step_count)
        self._device_steps = self.opt.step_count
        self.batch._offsets = None  # the count scans are part of the graph: they rerun for every loaded batch
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out, losses = self._eager()
        self.total_loss, self.logits_b = losses.total_loss, out.logits_b
        self._lr = (self.opt.param_groups[0]["lr"], self.opt.param_groups[0]["weight_decay"])

    def __call__(self):
        if self.graph is None or self._lr != (self.opt.param_groups[0]["lr"], self.opt.param_groups[0]["weight_decay"]):
            self._capture()  # (the capture itself executes nothing: replay below runs the first real step)
        if self._device_steps != self.opt.step_count:  # eager steps were taken in between
            self.opt.set_device_step(self.opt.step_count)
        self.graph.replay()
        self.model.engine().params_changed()  # the replayed optimizer launch rewrote theta
        self.opt.step_count += 1
        self._device_steps = self.opt.step_count
        return self.total_loss
