"""`train_artifact_model` (reference permutect/training/model_training.py:49-201) on the device engine.

Same structure as the reference: `num_epochs` training epochs (+ calibration epochs that only fit the calibration
parameters), every parent batch downsampled twice, balancer weights, the adversarial source strength ramp, loss
bookkeeping per (source, label, variant type, count bins), ReduceLROnPlateau on the epoch's mean loss, best-checkpoint
keep/rollback, a validation pass per epoch.  What differs is where the work happens: batches are composed on the device
from HBM-resident chunks, downsampling is two launches, forward / losses / backward / clip + AdamW are the fused kernels,
and the only host synchronisations are one per epoch (mean loss for the scheduler and the checkpoint decision).  Plots,
tensorboard and the worst-offender report of the reference are out of scope."""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from permutect_amd.data.reads_dataset import ReadsDataset
from permutect_amd.enums import Epoch
from permutect_amd.parameters import TrainingParameters
from permutect_amd.training.balancer import Balancer
from permutect_amd.training.distributed import BucketedGradAllReduce, assert_replicas_identical, rank0_decides
from permutect_amd.training.downsampler import Downsampler
from permutect_amd.training.loss_recorder import PRIMARY, LossRecorder, collect_evaluation_data
from permutect_amd.training.optimizer import FusedClipAdamW, backpropagate  # noqa: F401  (backpropagate: re-exported, reference misc_utils.py:125)


class PlateauScheduler:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', threshold_mode='rel') for an optimizer exposing param_groups
    (reference model_training.py:74-80: factor 0.2, patience 5, threshold 1e-3, min_lr = lr / 100)."""

    def __init__(self, optimizer, factor=0.2, patience=5, threshold=1e-3, min_lr=0.0):
        self.opt, self.factor, self.patience, self.threshold, self.min_lr = optimizer, factor, patience, threshold, min_lr
        self.best, self.bad = math.inf, 0

    def step(self, metric: float):
        if metric < self.best * (1 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            for g in self.opt.param_groups:
                g["lr"] = max(g["lr"] * self.factor, self.min_lr)
            self.bad = 0


class Checkpoint:
    """Keep the best model (by training mean loss) and roll back when the loss blows up (reference training/checkpoint.py)."""

    def __init__(self, model, optimizer):
        self.model, self.opt = model, optimizer
        self.best_loss, self.state, self.opt_state = math.inf, None, None

    def save_checkpoint_if_needed(self, epoch: int, loss: float):
        if loss < self.best_loss:
            self.best_loss = loss
            self.state = self.model.engine().space.theta.detach().clone()
            self.opt_state = self.opt.state_dict()

    def should_roll_back(self, loss: float) -> bool:
        return self.state is not None and (not math.isfinite(loss) or loss > 2 * self.best_loss)

    def load_checkpoint(self) -> bool:
        if self.state is None:
            return False
        with torch.no_grad():
            self.model.engine().space.theta.copy_(self.state)
        self.opt.load_state_dict(self.opt_state)  # moments, step count and learning rate (reference checkpoint.py:29-31)
        return True

    def load_checkpoint_if_needed(self, loss: float):
        return self.should_roll_back(loss) and self.load_checkpoint()



def check_for_nan(model):
    """Reference misc_utils.py:159-166: fail the run when a parameter's gradient holds a NaN or an Inf.  One reduction over the
    flat gradient buffer that every `.grad` aliases; the parameters are only named once something is wrong."""
    gtheta = model.engine().space.gtheta
    if bool(torch.isfinite(gtheta).all()):
        return
    for name, param in model.named_parameters():
        if param.grad is not None and not bool(torch.isfinite(param.grad).all()):
            print(f"Invalid gradient (NaN or Inf) found in parameter: {name}")
    raise AssertionError("invalid gradient (NaN or Inf)")

def training_params_fit_downsampler(training_params) -> bool:
    """The balance fit can be skipped (uniform mixture weights) with `fit_downsampler = False` on the parameters object --
    tests that train for a few steps do; the reference always fits."""
    return bool(getattr(training_params, "fit_downsampler", True))


def train_artifact_model(model, train_dataset: ReadsDataset, valid_dataset: Optional[ReadsDataset],
                         training_params: TrainingParameters, chunk_variants: Optional[int] = 1 << 18, seed: int = 0,
                         dist=None, log=print, fix_alt_gather: bool = False, evaluate_every_epoch: bool = True,
                         evaluations: Optional[list] = None):
    """`fix_alt_gather`: the reference's DownsampledBatch gathers the kept alt reads without the offset of the ref region
    (SURVEY 0.5b), so its training steps see ref rows in place of alt reads; False reproduces that, True gathers the
    intended rows."""
    device = model._device
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    # datasets that fit host memory are page-locked once: their chunks then go to HBM by DMA straight from the dataset (a dataset
    # that stays a memory map keeps the staged copies; PMT_PIN_DATASET=0 opts out).  bench.py's `loader` lines take the same path.
    for ds in (train_dataset, valid_dataset):
        if ds is not None and device.type == "cuda":
            ds.pin_memory_if_it_fits()
    num_sources = train_dataset.validate_sources()  # (reference :63)
    balancer = Balancer(num_sources=num_sources, device=device)
    downsampler = Downsampler(num_sources=num_sources)
    if training_params_fit_downsampler(training_params):
        # reference :58-60: fit the Beta-mixture weights so that the downsampled counts spread over the count bins
        # (deterministic, on the CPU: ~15 s of tiny einsums; identical on every rank, which all see the whole dataset's totals)
        downsampler.optimize_downsampling_balance(train_dataset.totals_slvra)
    downsampler = downsampler.to(device)
    model.reset_source_predictor(num_sources)
    opt = FusedClipAdamW(model, lr=training_params.learning_rate, weight_decay=training_params.weight_decay)
    scheduler = PlateauScheduler(opt, min_lr=training_params.learning_rate / 100)
    checkpoint = Checkpoint(model, opt)
    reduce_grads = None
    if dist is not None:
        # replicas start from rank 0's weights (reset_source_predictor drew fresh ones from every rank's own generator)
        dist.broadcast(model.engine().space.theta, src=0)
        model.engine().params_changed()
        reduce_grads = BucketedGradAllReduce()
        model.engine().grad_hook = reduce_grads  # early bucket: reduced on a side stream under the rest of the backward
    rng = np.random.default_rng(seed + rank)
    history = []
    evaluations = [] if evaluations is None else evaluations
    last_epoch = training_params.num_epochs + training_params.num_calibration_epochs
    step_seed = seed * 1_000_003 + rank

    for epoch in range(1, last_epoch + 1):
        is_calibration = epoch > training_params.num_epochs
        model.source_predictor.set_adversarial_strength((2 / (1 + math.exp(-0.1 * (epoch - 1)))) - 1)
        for epoch_type in (Epoch.TRAIN, Epoch.VALID):
            dataset = train_dataset if epoch_type == Epoch.TRAIN else valid_dataset
            if dataset is None:
                continue
            model.set_epoch_type(epoch_type)
            recorder = LossRecorder(device, num_sources)
            cal_opt = None
            if is_calibration and epoch_type == Epoch.TRAIN:  # only the calibration parameters move (reference :145-147)
                cal_opt = torch.optim.AdamW(model.calibration_parameters(), lr=opt.param_groups[0]["lr"],
                                            weight_decay=opt.param_groups[0]["weight_decay"])
            loader = dataset.device_loader(training_params.batch_size, device, chunk_variants=chunk_variants, rng=rng,
                                           shuffle=epoch_type == Epoch.TRAIN, rank=rank, world_size=world)
            # data parallel: every rank must take the same number of optimizer steps (one all-reduce each)
            max_batches = None
            if dist is not None and epoch_type == Epoch.TRAIN:
                max_batches = min(len(dataset.device_loader(training_params.batch_size, device, chunk_variants=chunk_variants,
                                                            shuffle=False, rank=r, world_size=world)) for r in range(world))
            for n_parent, parent in enumerate(loader):
                if max_batches is not None and n_parent >= max_batches:
                    break
                for _ in range(2):  # two independent downsamplings of every parent batch (reference :153)
                    step_seed += 1
                    batch = downsampler.downsample(parent, seed=step_seed, fix_alt_gather=fix_alt_gather)
                    with torch.set_grad_enabled(epoch_type == Epoch.TRAIN):
                        output = model.compute_batch_output(batch, balancer)
                        losses = model.compute_batch_losses(output, batch)
                    recorder.record(output, losses, batch)
                    if epoch_type == Epoch.TRAIN:
                        if cal_opt is not None:
                            cal_opt.zero_grad(set_to_none=True)
                            model.engine().space.gtheta.zero_()
                            model.engine().space.bind_grads()
                            losses.total_loss.backward()
                            if reduce_grads is not None:  # same order as the main step: backward -> reduce -> clip -> step
                                reduce_grads(model.engine().space.gtheta)
                            torch.nn.utils.clip_grad_norm_(model.calibration_parameters(), max_norm=1.0)
                            cal_opt.step()
                        else:
                            opt.zero_grad()
                            losses.total_loss.backward()
                            opt.step(pre_reduce=reduce_grads)
            check_for_nan(model)  # reference model_training.py:168: after EVERY epoch, validation epochs included (whose gradients are the last training step's)
            if dist is not None:
                recorder.all_reduce(dist)
            mean_loss = recorder.mean_loss(PRIMARY)  # the epoch's one host sync
            model.engine().check_join_fault()  # (a split read set whose workgroups did not meet in time: the epoch's numbers are wrong -> raise)
            history.append((epoch, epoch_type.name, mean_loss))
            log(f"epoch {epoch} {epoch_type.name}: mean semisupervised loss {mean_loss:.5f}, lr {opt.param_groups[0]['lr']:.2e}")
            if epoch_type == Epoch.VALID and evaluate_every_epoch:
                # reference :184-195: after every validation epoch, an evaluation pass over BOTH datasets, three downsamplings
                # of every parent batch (its plots are out of scope; the accuracies go to the log)
                ev = collect_evaluation_data(
                    model, balancer, downsampler,
                    train_dataset.device_loader(training_params.inference_batch_size, device, chunk_variants=chunk_variants, shuffle=False,
                                                rank=rank, world_size=world),
                    valid_dataset.device_loader(training_params.inference_batch_size, device, chunk_variants=chunk_variants, shuffle=False,
                                                rank=rank, world_size=world),
                    seed=seed * 1000 + epoch, fix_alt_gather=fix_alt_gather)
                if dist is not None:
                    ev.all_reduce(dist)  # every tally (histogram, call counts, logit sums) in one collective
                model.engine().check_join_fault()
                log(f"epoch {epoch} evaluation: accuracy train {ev.accuracy(0):.4f}, valid {ev.accuracy(1):.4f}")
                evaluations.append((epoch, ev.accuracy(0), ev.accuracy(1)))
            if epoch_type == Epoch.TRAIN:
                scheduler.step(mean_loss)
                if not is_calibration:
                    checkpoint.save_checkpoint_if_needed(epoch, mean_loss)
                    rollback = checkpoint.should_roll_back(mean_loss)
                    if dist is not None:  # the statistics were all-reduced, so the ranks agree; rank 0's word makes it certain
                        rollback = rank0_decides(rollback, device)
                    if rollback and checkpoint.load_checkpoint():
                        log(f"epoch {epoch}: loss diverged, restored the best checkpoint")
    if dist is not None:
        assert_replicas_identical(model.engine().space.theta)
    return history
