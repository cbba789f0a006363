"""Body of tests/test_dp_gpu.py::test_rccl_overlapped_reduction_on_one_card: the "nccl" backend (RCCL on ROCm) with a process
group of ONE rank on the box's one MI355X.  An all-reduce over one rank is the identity, so the hooked training step must land
where the un-hooked one does -- but it gets there through the real RCCL communicator, the side stream that carries the early
gradient bucket and both stream joins of BucketedGradAllReduce (force_active=True), none of which a gloo test touches."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path: str):
    from bench import synth_arrays
    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.data.batch import Batch
    from permutect_amd.parameters import P0_DIMS, p0_params
    from permutect_amd.training.distributed import BucketedGradAllReduce
    from permutect_amd.training.optimizer import FusedClipAdamW

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)  # RANK / WORLD_SIZE / MASTER_* from the environment (world size 1)
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    probe = torch.arange(1000, dtype=torch.float32, device=dev)
    dist.all_reduce(probe, op=dist.ReduceOp.SUM)  # the communicator is up: a SUM over one rank changes nothing
    assert torch.equal(probe.cpu(), torch.arange(1000, dtype=torch.float32))

    rng = np.random.default_rng(5)
    batches = [Batch.from_arrays(*synth_arrays(rng, 4096, "wgs"), pack=True).copy_to(dev) for _ in range(3)]

    def run(hooked: bool):
        torch.manual_seed(9)
        model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
        model.train(True)
        opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
        eng = model.engine()
        hook = None
        if hooked:
            hook = BucketedGradAllReduce(force_active=True)
            eng.grad_hook = hook
        losses = []
        for b in batches:
            opt.zero_grad()
            out = model.compute_batch_output(b)
            loss = model.compute_batch_losses(out, b).total_loss
            loss.backward()
            opt.step(pre_reduce=hook)
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        return eng.space.theta.detach().cpu().clone(), losses, hook, eng.space.late_start

    plain, plain_losses, _, late_start = run(False)
    again, _, _, _ = run(False)
    hooked, hooked_losses, hook, _ = run(True)
    torch.save({"plain": plain, "again": again, "hooked": hooked, "plain_losses": plain_losses, "hooked_losses": hooked_losses,
                "early_reductions": hook.early_reductions, "side_stream": hook._side is not None, "pending": hook._pending is not None,
                "late_start": late_start, "backend": dist.get_backend()}, out_path)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
