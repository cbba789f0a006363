"""Development aid: host profile of the evaluation pass (collect_evaluation_data) over a synthetic 2^19-variant dataset."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.data.reads_dataset import ReadsDataset
from permutect_amd.parameters import P0_DIMS, p0_params
from permutect_amd.training.balancer import Balancer
from permutect_amd.training.downsampler import Downsampler
from permutect_amd.training.loss_recorder import collect_evaluation_data

dev = torch.device("cuda")
n = 1 << 19
ds = ReadsDataset(MemoryMappedData.from_arrays(*synth_arrays(np.random.default_rng(5050), n, "wgs")))
ds.pin_memory_if_it_fits()
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
bal, down = Balancer(1, dev), Downsampler(1).to(dev)


def one_pass():
    ev = collect_evaluation_data(model, bal, down, ds.device_loader(65536, dev, chunk_variants=1 << 18, shuffle=False), None, seed=3)
    return ev.accuracy(0)


one_pass()
torch.cuda.synchronize()
t = time.perf_counter()
one_pass()
print(f"{1e3 * (time.perf_counter() - t) / (3 * n // 65536):.3f} ms per evaluation step")
pr = cProfile.Profile()
pr.enable()
one_pass()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
