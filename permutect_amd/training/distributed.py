"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The reference has no distributed code at all; the semantics implemented here are fixed in
SURVEY.md section 8e:
  * filter_variants: the candidate index range is cut into `world` contiguous shards; no collective.
  * training: every rank runs forward/backward on its own batch; the flat gradient buffer (about 240 KB at P0) is
    all-reduced (SUM) in TWO buckets, then the identical clip + AdamW runs on every rank.  SUM, not mean, because the
    reference's total_loss is a batch sum (artifact_model.py:90): N ranks with batch B are exactly one process with
    batch N*B.  The buffer is laid out [early | late] (engine/plan.py: ParamSpace): the early bucket (every leaf the
    read-set backward kernel and the adversaries' row kernels own, ~93 % of the parameters) is final as soon as
    pmt_backward has been enqueued and is reduced on a side stream UNDERNEATH the kernels autograd still has to run
    (haplotype-CNN backward ~1 ms, info-MLP backward, parametrization adjoint); the late bucket follows on the main
    stream.  Order per step (reference misc_utils.py:125-129): backward -> reduce -> clip -> step.
  * per-epoch decisions (lr scheduler input, checkpoint rollback) use all-reduced loss statistics, decided on rank 0 and
    broadcast.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, end) of rank's share of n items (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class GradAllReduce:
    """`FusedClipAdamW.step(pre_reduce=GradAllReduce(group))`: SUM all-reduce of the flat gradient before the clip."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group

    def __call__(self, flat_grad: torch.Tensor) -> None:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)


class BucketedGradAllReduce:
    """The overlapped form of `GradAllReduce` for a flat buffer laid out [early | late]:

        hook = BucketedGradAllReduce(group);  engine.grad_hook = hook        # ReadSetEngine.backward calls hook.early(...)
        optimizer.step(pre_reduce=hook)                                      # reduces the rest and joins

    `early(flat, late_start)` is called right after the read-set backward kernel has been enqueued: flat[:late_start] is
    final from that point of the stream on, and its all-reduce is issued on a side stream that waits for exactly that
    point.  `__call__(flat)` (the optimizer's pre-reduce hook, after loss.backward() has returned) reduces
    flat[late_start:] and makes the current stream wait for both.  The sum of the two bucket reductions is the flat SUM
    all-reduce (tests/test_distributed_cpu.py).  Without a call to `early` since the last step (no read-set backward ran,
    e.g. an empty batch) the whole buffer is reduced in one piece.  Works on CPU tensors (gloo) without streams.

    Contract: ONE backward per optimizer step.  A second `early` while a reduction is pending would add a second backward's
    local gradients into a bucket that is being (or has been) summed over the ranks -- a data race and a wrong sum -- so it
    raises; gradient accumulation over several backwards has to use the flat `GradAllReduce` at the step instead.

    `force_active`: run the collectives even in a process group of ONE rank (an all-reduce over one rank is the identity):
    the RCCL communicator, the side stream and the stream joins of the overlapped path can then be exercised on a single
    card (tests/test_dp_gpu.py)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, force_active: bool = False):
        self.group = group
        self.force_active = force_active
        self._pending = None   # (work, late_start)
        self._side = None
        self.early_reductions = 0  # how many early buckets went out on the side stream (tests; bench.py's `collective` record)
        self.steps = 0             # optimizer steps reduced through __call__
        self.early_bytes = 0       # bytes of the LAST step's early / late bucket (what bench.py reports)
        self.late_bytes = 0

    def _active(self) -> bool:
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return self.force_active or dist.get_world_size(self.group) > 1

    def describe(self) -> dict:
        """What the collectives saw, for bench.py's N > 1 lines: the backend and world size as torch.distributed reports them,
        the early buckets that went out per optimizer step, the bytes of the last step's two buckets."""
        on = dist.is_available() and dist.is_initialized()
        return {"backend": dist.get_backend(self.group) if on else None, "world_size": dist.get_world_size(self.group) if on else 1,
                "steps_reduced": self.steps, "early_buckets_per_step": (self.early_reductions / self.steps) if self.steps else 0.0,
                "early_bytes": self.early_bytes, "late_bytes": self.late_bytes, "op": "SUM",
                "early_bucket_on_side_stream": self._side is not None}

    def early(self, flat_grad: torch.Tensor, late_start: int) -> None:
        if not self._active() or late_start <= 0:
            return
        if self._pending is not None:
            raise RuntimeError("BucketedGradAllReduce: a second backward before the optimizer step (its early bucket is already "
                               "being reduced); reduce accumulated gradients with GradAllReduce at the step instead")
        bucket = flat_grad[:late_start]
        if flat_grad.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(flat_grad.device)
            self._side.wait_stream(torch.cuda.current_stream(flat_grad.device))  # the read-set gradients are final here
            with torch.cuda.stream(self._side):
                work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending = (work, late_start)
        self.early_reductions += 1
        self.early_bytes = bucket.numel() * bucket.element_size()

    def reset(self) -> None:
        """Drop a reduction left pending by a step that was abandoned between loss.backward() and optimizer.step() (an exception
        caught by the caller, a gradient probe with the hook installed): waits for the early bucket's collective -- every rank
        issued it, so it completes -- and forgets it, after which the next backward may issue a new one.  FusedClipAdamW.zero_grad
        calls this, so an abandoned step cannot poison the hook for good.  NOTE the raise in `early` is rank-local: a rank that
        raises there has left the other ranks inside their all-reduce; callers that catch it must tear the process group down (the
        training loop does not catch it)."""
        if self._pending is None:
            return
        work, _ = self._pending
        self._pending = None
        work.wait()
        if self._side is not None:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)

    def __call__(self, flat_grad: torch.Tensor) -> None:
        if not self._active():
            return
        self.steps += 1
        if self._pending is None:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            self.early_bytes, self.late_bytes = 0, flat_grad.numel() * flat_grad.element_size()
            return
        work, late_start = self._pending
        self._pending = None
        self.late_bytes = (flat_grad.numel() - late_start) * flat_grad.element_size()
        if late_start < flat_grad.numel():
            dist.all_reduce(flat_grad[late_start:], op=dist.ReduceOp.SUM, group=self.group)
        work.wait()  # CUDA: the current stream waits for the side-stream reduction; CPU: blocks
        if flat_grad.is_cuda:
            torch.cuda.current_stream(flat_grad.device).wait_stream(self._side)


_HOST_GROUP = None


def host_group():
    """A gloo process group over the ranks of the default one, for moving HOST arrays between ranks (the filter's per-shard result rows
    to rank 0): RCCL moves device memory only.  Created once, collectively -- every rank must reach the first call together; None
    (= the default group) when the default backend is gloo already."""
    global _HOST_GROUP
    if dist.get_backend() == "gloo":
        return None
    if _HOST_GROUP is None:
        _HOST_GROUP = dist.new_group(backend="gloo")
    return _HOST_GROUP


def init_from_env(device_type: str = "cuda"):
    """What the command-line tools call FIRST, before anything touches the GPU: under torchrun (WORLD_SIZE > 1 in the environment; one
    process per GPU) pick this rank's card by LOCAL_RANK, then join the job's process group -- backend "nccl" (= RCCL over xGMI on
    ROCm) for device tensors, gloo without a GPU.  Returns (torch.distributed or None, rank, world_size, device).  MASTER_ADDR /
    MASTER_PORT come from the launcher (use 127.0.0.1 on one node)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cuda = device_type == "cuda" and torch.cuda.device_count() > 0  # (device_count does not initialise the GPU)
    if world <= 1:
        return None, 0, 1, torch.device("cuda", torch.cuda.current_device()) if cuda else torch.device("cpu")
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if cuda:
        torch.cuda.set_device(local % torch.cuda.device_count())
    if not dist.is_initialized():
        # (PMT_DIST_BACKEND=gloo: a rehearsal of N ranks on ONE card -- RCCL wants a device per rank; tests/test_dp_gpu.py)
        backend = os.environ.get("PMT_DIST_BACKEND") or ("nccl" if cuda else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist, rank, world, torch.device("cuda", torch.cuda.current_device()) if cuda else torch.device("cpu")


def assert_replicas_identical(flat: torch.Tensor, what: str = "parameters", group: Optional[dist.ProcessGroup] = None) -> None:
    """Data parallel rests on every rank taking the identical step: the element-wise MAX and MIN of a buffer over the ranks must be the
    same bits.  Two small all-reduces; the training loop calls it once, at the end (a replica that drifted means every step since was
    a different model on each card: fail loudly rather than save rank 0's)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    hi, lo = flat.detach().clone(), flat.detach().clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    if not torch.equal(hi, lo):
        raise RuntimeError(f"data-parallel replicas diverged: {int((hi != lo).sum())} of {flat.numel()} {what} differ between ranks")


def all_reduce_sum_(t: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def rank0_decides(flag: bool, device: torch.device, group: Optional[dist.ProcessGroup] = None) -> bool:
    """Rank 0's boolean, on every rank (checkpoint save / rollback decisions must be identical everywhere)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return flag
    t = torch.tensor([1 if flag else 0], device=device, dtype=torch.int32)
    dist.broadcast(t, src=0, group=group)
    return bool(t.item())
