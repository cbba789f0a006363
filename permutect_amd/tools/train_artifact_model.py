"""`train_artifact_model` on the MI355X engine: the reference's command line (tools/train_artifact_model.py:22-88) with the
reference's flag names -- `main_without_parsing(args)` takes the same Namespace the reference's own tool test builds
(test/tools/test_train_permutect_model.py:14-64).  Thin by design: load the tar, nine folds to train on and the tenth to validate
(reference :34-38), build (or load) the model, `training.model_training.train_artifact_model`, `save_model`.  Tensorboard output and
the plots are out of scope: `--tensorboard_dir` is accepted and ignored; the epoch losses go to stdout.

    python -m permutect_amd.tools.train_artifact_model --train_tar data.tar --output model.pt --read_layers 30 -2 -2 -2 ...
"""
from __future__ import annotations

import argparse

import torch

from permutect_amd import constants
from permutect_amd.architecture.artifact_model import ArtifactModel, load_model
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.data.reads_dataset import ReadsDataset, all_but_last_fold, last_fold_only
from permutect_amd.parameters import (add_model_params_to_parser, add_training_params_to_parser, parse_model_params,
                                      parse_training_params)
from permutect_amd.training.distributed import init_from_env
from permutect_amd.training.model_training import train_artifact_model

NUM_FOLDS = 10  # reference :33


def main_without_parsing(args, log=print):
    """Under torchrun (`python -m torch.distributed.run --nproc-per-node N -m permutect_amd.tools.train_artifact_model ...`: WORLD_SIZE > 1)
    the run is data parallel, one process per GPU: this rank's card and the process group first (training/distributed.py:
    init_from_env), every rank trains on its contiguous shard of the folds with the flat gradient summed over RCCL each step
    (training/model_training.py), rank 0 writes the model -- the replicas are identical."""
    params, training_params = parse_model_params(args), parse_training_params(args)
    if torch.cuda.device_count() == 0:
        raise RuntimeError("permutect_amd trains on an MI355X (ROCm device 'cuda'); there is no CPU path")
    dist, rank, world, device = init_from_env()  # before anything else touches the GPU
    log = log if rank == 0 else (lambda *a, **k: None)
    pretrained = getattr(args, constants.PRETRAINED_ARTIFACT_MODEL_NAME, None)
    data = MemoryMappedData.load_from_tarfile(getattr(args, constants.TRAIN_TAR_NAME))
    train_dataset = ReadsDataset(data, num_folds=NUM_FOLDS, folds_to_use=all_but_last_fold(NUM_FOLDS))
    valid_dataset = ReadsDataset(data, num_folds=NUM_FOLDS, folds_to_use=last_fold_only(NUM_FOLDS))
    if pretrained is not None:
        model, _, _ = load_model(pretrained, device=device)
    else:
        model = ArtifactModel(params=params, num_read_features=train_dataset.num_read_features(),
                              num_info_features=train_dataset.num_info_features(), haplotypes_length=train_dataset.haplotypes_length(),
                              device=device)
    history = train_artifact_model(model, train_dataset, valid_dataset, training_params, dist=dist, log=log)
    if rank == 0:
        model.save_model(path=getattr(args, constants.OUTPUT_NAME))
    if dist is not None:
        dist.barrier()  # (nobody leaves -- and tears the process group down -- while rank 0 still writes)
    return history


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description="train the Permutect artifact model on an MI355X")
    add_model_params_to_parser(parser)
    add_training_params_to_parser(parser)
    parser.add_argument("--" + constants.TRAIN_TAR_NAME, type=str, required=True, help="dataset tar produced by the reference's preprocess_dataset")
    parser.add_argument("--" + constants.OUTPUT_NAME, type=str, required=True, help="output artifact model file (.pt, the reference's format)")
    parser.add_argument("--" + constants.TENSORBOARD_DIR_NAME, type=str, default="tensorboard", required=False, help="accepted and ignored")
    return parser.parse_args(argv)


def main():
    main_without_parsing(parse_arguments())


if __name__ == "__main__":
    main()
