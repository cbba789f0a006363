"""Fused global-norm clip + AdamW over the model's flat parameter buffer (pmt_clip_adamw), and the
`backpropagate` helper with the reference's call signature (reference permutect/misc_utils.py:125-129;
optimizer construction reference training/model_training.py:68-72)."""
from __future__ import annotations

import ctypes as C

import torch

from permutect_amd.engine import lib as L


class FusedClipAdamW:
    """AdamW (decoupled weight decay, bias correction, torch defaults betas=(0.9, 0.999), eps=1e-8) preceded by
    clip_grad_norm_(max_norm) over ALL parameters, as two HIP launches on the flat buffers, with no host sync."""

    def __init__(self, model, lr: float = 1e-3, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: float = 1.0):
        self.model = model
        self.lr, self.weight_decay, self.betas, self.eps, self.max_grad_norm = lr, weight_decay, betas, eps, max_grad_norm
        self.step_count = 0
        self._space = None
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay}]  # lr schedulers poke this

    def _bind(self):
        space = self.model.engine().space
        if space is not self._space:  # first use, or the model re-flattened (reset_source_predictor)
            self._space = space
            self.exp_avg = torch.zeros_like(space.theta)
            self.exp_avg_sq = torch.zeros_like(space.theta)
            self.scratch = torch.zeros(1024, dtype=torch.float32, device=space.theta.device)
            self.grad_norm = torch.zeros(1, dtype=torch.float32, device=space.theta.device)
            self.step_count = 0
        return space

    def zero_grad(self, set_to_none: bool = True):
        space = self._bind()
        hook = getattr(getattr(self.model, "_engine", None), "grad_hook", None)
        if hook is not None and hasattr(hook, "reset"):
            hook.reset()  # (a step abandoned between backward and step left its early bucket pending: training/distributed.py)
        space.gtheta.zero_()
        space.bind_grads()

    def step(self, pre_reduce=None, step_on_device: bool = False):
        """`pre_reduce(flat_grad)` (optional) runs before the clip: the data-parallel all-reduce hook.
        `step_on_device`: the step number (Adam's bias correction) is read from, and incremented in, device memory by the
        launch itself (PMT_STEP_ON_DEVICE) -- what a captured graph needs; the caller keeps `step_count` in step
        (tests/graph_capture.py)."""
        space = self._bind()
        if pre_reduce is not None:
            pre_reduce(space.gtheta)
        if not step_on_device:
            self.step_count += 1
        hp = L.PmtAdamW(self.param_groups[0]["lr"], self.betas[0], self.betas[1], self.eps,
                        self.param_groups[0]["weight_decay"], self.max_grad_norm,
                        L.STEP_ON_DEVICE if step_on_device else self.step_count, 0)
        L.check(L.load().pmt_clip_adamw(space.theta.data_ptr(), space.gtheta.data_ptr(), self.exp_avg.data_ptr(),
                                        self.exp_avg_sq.data_ptr(), space.size, C.byref(hp), self.scratch.data_ptr(),
                                        self.grad_norm.data_ptr(), L.raw_stream()),
                "pmt_clip_adamw")
        eng = getattr(self.model, "_engine", None)
        if eng is not None:
            eng.params_changed()  # theta was written through a raw pointer: packed weights / phi of an earlier forward are stale

    def set_device_step(self, n: int):
        """the device-resident step counter of `step(step_on_device=True)`: n steps taken so far"""
        self._bind()
        self.scratch.view(torch.int32)[L.STEP_SLOT] = n

    def state_dict(self):
        self._bind()
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [dict(g) for g in self.param_groups]}  # torch optimizers keep lr / weight decay here too

    def load_state_dict(self, state):
        self._bind()
        self.step_count = state["step"]
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])
        for g, saved in zip(self.param_groups, state.get("param_groups", [])):
            g.update(saved)


def backpropagate(optimizer, loss: torch.Tensor, params_to_clip=()):
    """zero_grad -> backward -> clip_grad_norm_(1.0) -> step, like the reference helper.  With a FusedClipAdamW the
    clip is part of the fused step; with a stock torch optimizer the reference sequence is executed as is."""
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if isinstance(optimizer, FusedClipAdamW):
        optimizer.step()
    else:
        torch.nn.utils.clip_grad_norm_(params_to_clip, max_norm=1.0)
        optimizer.step()
