"""The artifact-model stage of `filter_variants` on the MI355X engine (reference tools/filter_variants.py:292-320, :350-357:
`generate_posterior_data` + `MemoryMappedData.from_generator`): every candidate of a dataset tar goes through the model's forward and
comes out as a read-less Datum row whose info array is its embedding and whose CACHED_ARTIFACT_LOGIT is its logit -- the input of
the posterior model.  What stands in front of this stage in the reference (plain-text parsing, VCF annotation) and behind it (the
posterior model, the filtered VCF) is out of scope: the tool reads a dataset tar in the reference's format and writes the
posterior data as a tar in the same format.

    python -m permutect_amd.tools.filter_variants --test_dataset_tar candidates.tar --artifact_model model.pt --output posterior.tar

Under torchrun (`python -m torch.distributed.run --nproc-per-node N -m permutect_amd.tools.filter_variants ...`: WORLD_SIZE > 1) the
candidates are cut into N contiguous shards, one process per GPU, no collective on the data path; rank 0 concatenates the shards'
rows in dataset order and writes the tar (tools/posterior_data.py: make_posterior_mmap)."""
from __future__ import annotations

import argparse
import time

import torch

from permutect_amd import constants
from permutect_amd.architecture.artifact_model import load_model
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.data.reads_dataset import ReadsDataset
from permutect_amd.tools.posterior_data import make_posterior_mmap
from permutect_amd.training.distributed import init_from_env

TEST_DATASET_TAR_NAME = "test_dataset_tar"
ARTIFACT_MODEL_NAME = "artifact_model"  # (reference constants.py: ARTIFACT_MODEL_NAME)
DEFAULT_BATCH_SIZE = 65536  # the reference's flag defaults to 64 (tools/filter_variants.py:81); the rows do not depend on it


def main_without_parsing(args, log=print):
    if torch.cuda.device_count() == 0:
        raise RuntimeError("permutect_amd filters on an MI355X (ROCm device 'cuda'); there is no CPU path")
    dist, rank, world, device = init_from_env()  # this rank's card and the process group, before anything else touches the GPU
    model, _, _ = load_model(getattr(args, ARTIFACT_MODEL_NAME), device=device)
    data = MemoryMappedData.load_from_tarfile(getattr(args, TEST_DATASET_TAR_NAME))
    dataset = ReadsDataset(data)
    t0 = time.perf_counter()
    posterior = make_posterior_mmap(dataset, model, getattr(args, constants.BATCH_SIZE_NAME), device=device,
                                    chunk_variants=getattr(args, "chunk_variants", None), rank=rank, world_size=world)
    if rank == 0:
        dt = time.perf_counter() - t0
        log(f"{len(posterior)} candidates through the artifact model on {world} GPU(s) in {dt:.3f} s ({len(posterior) / max(dt, 1e-9) / 1e6:.2f} M/s, "
            "disk to posterior rows)")
        posterior.save_to_tarfile(getattr(args, constants.OUTPUT_NAME))
    if dist is not None:
        dist.barrier()
    return posterior


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description="the artifact-model stage of filter_variants on an MI355X: candidates -> posterior data")
    parser.add_argument("--" + TEST_DATASET_TAR_NAME, type=str, required=True, help="dataset tar (the reference's format) of the candidates")
    parser.add_argument("--" + ARTIFACT_MODEL_NAME, type=str, required=True, help="artifact model from train_artifact_model (.pt, the reference's format)")
    parser.add_argument("--" + constants.OUTPUT_NAME, type=str, required=True, help="output tar of the posterior data")
    parser.add_argument("--" + constants.BATCH_SIZE_NAME, type=int, default=DEFAULT_BATCH_SIZE, required=False, help="batch size")
    parser.add_argument("--chunk_variants", type=int, default=None, required=False, help="candidates per HBM-resident chunk (default: the loader's)")
    return parser.parse_args(argv)


def main():
    main_without_parsing(parse_arguments())


if __name__ == "__main__":
    main()
