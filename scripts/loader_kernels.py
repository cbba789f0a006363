#!/usr/bin/env python3
"""Development aid: device time per kernel of the filter forward when its batches come through the device chunk loader, next to the
same forward on a resident batch -- what composing a batch on the device and reading rows through the gather index cost."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.batch import Batch  # noqa: E402
from permutect_amd.data.memory_mapped_data import MemoryMappedData  # noqa: E402
from permutect_amd.data.reads_dataset import ReadsDataset  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ints, floats, packed = synth_arrays(rng, 1 << 20, "wgs")
ds = ReadsDataset(MemoryMappedData.from_arrays(ints, floats, packed)).pin_memory()
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.engine()
model.train(False)
resident = Batch.from_arrays(*synth_arrays(rng, 65536, "wgs"), pack=True).copy_to(dev)


def run(kind):
    n = 0
    with torch.inference_mode():
        if kind == "loader":
            for cb in ds.device_loader(65536, dev, chunk_variants=1 << 18, shuffle=False):
                model.compute_batch_output(cb)
                n += 1
        else:
            for _ in range(16):
                model.compute_batch_output(resident)
                n += 1
    torch.cuda.synchronize()
    return n


for kind in ("resident", "loader"):
    run(kind)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        n = run(kind)
    rows = sorted(((e.key, e.device_time_total / n, e.count / n) for e in prof.key_averages() if e.device_time_total > 0), key=lambda r: -r[1])
    print(f"== {kind}: {sum(r[1] for r in rows):.0f} us of device time per batch")
    for k, t, c in rows[:14]:
        print(f"{t:9.1f} us  x{c:4.1f}  {k[:100]}")
