"""Development aid: the epoch loop the CLI runs (training/model_training.train_artifact_model) on a synthetic 2^20-variant dataset, for
`rocprofv3 --kernel-trace --stats -- python3 scripts/epoch_profile.py` (what a loop step launches beside the bench's resident step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.data.reads_dataset import ReadsDataset
from permutect_amd.parameters import P0_DIMS, TrainingParameters, p0_params
from permutect_amd.training.model_training import train_artifact_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda")
ds = ReadsDataset(MemoryMappedData.from_arrays(*synth_arrays(np.random.default_rng(5050), n, "wgs")))
ds.pin_memory_if_it_fits()
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
stamps = []
train_artifact_model(model, ds, None, TrainingParameters(batch_size=65536, num_epochs=epochs, learning_rate=1e-3, weight_decay=0.01, fit_downsampler=False),
                     chunk_variants=1 << 18, seed=9, log=lambda m: (stamps.append(time.perf_counter()), print(m, flush=True)), evaluate_every_epoch=False)
steps = 2 * (-(-n // 65536))
print("epoch seconds", [b - a for a, b in zip(stamps, stamps[1:])], "ms per optimizer step", [1e3 * (b - a) / steps for a, b in zip(stamps, stamps[1:])])
