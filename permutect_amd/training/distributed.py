"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The reference has no distributed code at all; the semantics implemented here are fixed in
SURVEY.md section 8e:
  * filter_variants: the candidate index range is cut into `world` contiguous shards; no collective.
  * training: every rank runs forward/backward on its own batch, then ONE all-reduce (SUM) of the flat gradient buffer
    (about 240 KB at P0: latency-bound, so a single bucket), then the identical clip + AdamW on every rank.  SUM, not
    mean, because the reference's total_loss is a batch sum (artifact_model.py:90): N ranks with batch B are exactly one
    process with batch N*B.
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
