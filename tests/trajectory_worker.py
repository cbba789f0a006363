"""Body of tests/test_trajectory_gpu.py: N optimizer steps of the HIP training path (forward, losses, backward, fused clip + AdamW)
on a fixed sequence of seeded batches; writes the loss of every step and the final parameters.  Run as a subprocess so that the
test can point PMT_LIB at another build of the library (the six-MFMA backward) and compare the two against one oracle run."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STEPS, BATCH, NBATCH, LR, WD = 100, 512, 8, 1e-3, 0.01


def make_batches():
    from bench import synth_arrays
    rng = np.random.default_rng(2024)
    return [synth_arrays(rng, BATCH, "wgs") for _ in range(NBATCH)]


def initial_state_dict():
    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.parameters import P0_DIMS, p0_params
    torch.manual_seed(31)
    model = ArtifactModel(p0_params(), device=torch.device("cpu"), **P0_DIMS)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def main(out_path: str):
    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.data.batch import Batch
    from permutect_amd.parameters import P0_DIMS, p0_params
    from permutect_amd.training.optimizer import FusedClipAdamW
    dev = torch.device("cuda:0")
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    model.load_state_dict(initial_state_dict())
    model.train(True)
    opt = FusedClipAdamW(model, lr=LR, weight_decay=WD)
    batches = [Batch.from_arrays(*a).copy_to(dev) for a in make_batches()]
    losses = []
    for step in range(STEPS):
        b = batches[step % NBATCH]
        opt.zero_grad()
        loss = model.compute_batch_losses(model.compute_batch_output(b), b).total_loss
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    torch.save({"losses": torch.stack(losses).cpu().double().numpy(),
                "params": {n: p.detach().cpu().clone() for n, p in model.named_parameters()},
                "lib": os.environ.get("PMT_LIB", "default")}, out_path)


if __name__ == "__main__":
    main(sys.argv[1])
