"""The captured-graph training step (tests/graph_capture.py: test infrastructure that proves the C ABI capturable) against the eager step on the same sequence of batches: same kernels,
so parameters agree to the order-dependence of the float atomics; the batch data, its group count and the optimizer's step
number all reach the replay through device memory."""
import numpy as np
import pytest
import torch

from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.engine import lib as L
from tests.graph_capture import GraphedTrainStep, StaticBatch
from permutect_amd.parameters import P0_DIMS, p0_params
from permutect_amd.training.optimizer import FusedClipAdamW, backpropagate

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _host_batch(nb, seed):
    from bench import synth_arrays
    return Batch.from_arrays(*synth_arrays(np.random.default_rng(seed), nb, "wgs"), pack=True)


@pytest.mark.parametrize("nb", [64, 1024])
def test_graph_replay_matches_eager_steps(nb):
    batches = [_host_batch(nb, 50 + i) for i in range(6)]
    assert len({b.plan().num_groups for b in batches}) > 1 or nb == 64  # the group count really varies from batch to batch
    torch.manual_seed(3)
    eager = ArtifactModel(p0_params(), device=DEV, **P0_DIMS)
    torch.manual_seed(3)
    graphed = ArtifactModel(p0_params(), device=DEV, **P0_DIMS)
    eager.train(True)
    opt_e = FusedClipAdamW(eager, lr=1e-3, weight_decay=0.01)
    opt_g = FusedClipAdamW(graphed, lr=1e-3, weight_decay=0.01)
    static = StaticBatch(nb, max_reads=26 * nb, device=DEV, int_cols=58, float_cols=77)
    step = GraphedTrainStep(graphed, opt_g, static)
    losses_e, losses_g = [], []
    for b in batches:
        out = eager.compute_batch_output(b.copy_to(DEV))
        le = eager.compute_batch_losses(out, b.copy_to(DEV)).total_loss
        backpropagate(opt_e, le, params_to_clip=eager.parameters())
        losses_e.append(float(le.detach()))
        static.load(b)
        losses_g.append(float(step().detach()))
    torch.cuda.synchronize()
    assert opt_g.step_count == opt_e.step_count == len(batches)
    np.testing.assert_allclose(losses_g, losses_e, rtol=2e-4)      # (each later loss depends on the earlier updates)
    te, tg = eager.engine().space.theta, graphed.engine().space.theta
    # parameters travel in different orders inside the two flat buffers only if the models differ: same class, same order
    assert float((te - tg).abs().max()) <= 2e-5, float((te - tg).abs().max())
    assert float((opt_e.exp_avg_sq - opt_g.exp_avg_sq).abs().max()) <= 1e-4 * float(opt_e.exp_avg_sq.abs().max())


def test_loads_and_replays_without_synchronising_match_eager_steps():
    """bench.py's loop: pinned host batches, load() + replay back to back with no synchronisation, so the host runs several
    batches ahead of the device (more loads than StaticBatch has staging slots).  Every replay must have seen ITS batch's
    group plan: the per-step losses equal the eager steps' on the same sequence."""
    nb = 1024
    batches = [_host_batch(nb, 70 + i) for i in range(10)]
    assert len({b.plan().num_groups for b in batches}) > 1
    torch.manual_seed(5)
    eager = ArtifactModel(p0_params(), device=DEV, **P0_DIMS)
    torch.manual_seed(5)
    graphed = ArtifactModel(p0_params(), device=DEV, **P0_DIMS)
    eager.train(True)
    opt_e = FusedClipAdamW(eager, lr=1e-3, weight_decay=0.01)
    opt_g = FusedClipAdamW(graphed, lr=1e-3, weight_decay=0.01)
    losses_e = []
    for b in batches:
        d = b.copy_to(DEV)
        le = eager.compute_batch_losses(eager.compute_batch_output(d), d).total_loss
        backpropagate(opt_e, le, params_to_clip=eager.parameters())
        losses_e.append(float(le.detach()))
    static = StaticBatch(nb, max_reads=26 * nb, device=DEV, int_cols=58, float_cols=77)
    step = GraphedTrainStep(graphed, opt_g, static)
    pinned = [b.pin_memory() for b in batches]
    static.load(pinned[0])
    step()  # capture (and the first real step) -- then put the model back and run the whole sequence without a sync
    torch.cuda.synchronize()
    torch.manual_seed(5)
    fresh = ArtifactModel(p0_params(), device=DEV, **P0_DIMS)
    with torch.no_grad():
        graphed.engine().space.theta.copy_(fresh.engine().space.theta)
        opt_g.exp_avg.zero_(); opt_g.exp_avg_sq.zero_()
    opt_g.step_count = 0
    on_device = []
    for b in pinned:
        static.load(b)
        on_device.append(step().detach().clone())  # (no .item(): nothing waits for the device inside the loop)
    torch.cuda.synchronize()
    np.testing.assert_allclose([float(t) for t in on_device], losses_e, rtol=2e-4)
    te, tg = eager.engine().space.theta, graphed.engine().space.theta
    assert float((te - tg).abs().max()) <= 2e-5, float((te - tg).abs().max())


def test_static_batch_refuses_what_it_cannot_hold():
    static = StaticBatch(64, max_reads=500, device=DEV, int_cols=58, float_cols=77)
    with pytest.raises(L.PmtError, match="capacity"):
        static.load(_host_batch(64, 1))          # ~830 reads
    with pytest.raises(L.PmtError):
        static.load(_host_batch(32, 1))          # another batch size
