"""Rank body of tests/test_dp_gpu.py: `train_artifact_model` under torch.distributed, two ranks on one MI355X (gloo:
the box has one card, and RCCL wants a device per rank; the product code is the same, only the backend string differs).
Launched by torch.distributed.run; writes each rank's parameters and history for the test to compare."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir: str):
    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.data.memory_mapped_data import MemoryMappedData
    from permutect_amd.data.reads_dataset import ReadsDataset, all_but_last_fold, last_fold_only
    from permutect_amd.parameters import P0_DIMS, TrainingParameters, p0_params
    from permutect_amd.training.model_training import train_artifact_model

    dist.init_process_group("gloo")
    rank = dist.get_rank()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    mm = MemoryMappedData.load_from_tarfile(os.path.join(ROOT, "tests", "golden", "tiny_dataset.tar"))
    train = ReadsDataset(mm, num_folds=5, folds_to_use=all_but_last_fold(5))
    valid = ReadsDataset(mm, num_folds=5, folds_to_use=last_fold_only(5))
    torch.manual_seed(100 + rank)  # DIFFERENT initial weights per rank: the loop must make the replicas identical itself
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    hist = train_artifact_model(model, train, valid, TrainingParameters(batch_size=8, num_epochs=2, num_calibration_epochs=1,
                                                                        learning_rate=1e-3, fit_downsampler=False),
                                chunk_variants=None, seed=3, dist=dist, log=lambda *_: None)
    eng = model.engine()
    torch.cuda.synchronize()
    torch.save({"theta": eng.space.theta.detach().cpu(), "history": hist, "late_start": eng.space.late_start,
                "hook": type(eng.grad_hook).__name__}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
