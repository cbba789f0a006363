#!/usr/bin/env python3
"""Does the per-variant part of the forward (info MLP + haplotype CNN) of batch i + 1 overlap with the read-set kernel of
batch i when they run on two streams?  (development aid)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.batch import Batch  # noqa: E402
from permutect_amd.engine.runtime import PhiFunction, ReadSetFunction  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402

dev = torch.device("cuda:0")
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(False)
rng = np.random.default_rng(0)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
batches = [Batch.from_arrays(*synth_arrays(rng, b, "wgs")).copy_to(dev) for _ in range(4)]
for x in batches:
    x.plan(allow_split=True)
eng = model.engine()
n = 40
with torch.inference_mode():
    prog = eng.plan.phi_program(model)
    phi = PhiFunction.apply(eng, prog, eng.trigger)
    eng.pack(phi.detach().contiguous())
    for x in batches:
        model.compute_batch_output(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        ve = model.variant_embedding(batches[i % 4])
        ReadSetFunction.apply(eng, batches[i % 4], phi, ve)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"sequential: {1e3 * (t1 - t0) / n:.3f} ms/batch")
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    events = [torch.cuda.Event() for _ in range(n + 1)]
    ves = [None] * (n + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        ves[0] = model.variant_embedding(batches[0])
        events[0].record(side)
    for i in range(n):
        with torch.cuda.stream(side):
            ves[i + 1] = model.variant_embedding(batches[(i + 1) % 4])
            events[i + 1].record(side)
        main.wait_event(events[i])
        ReadSetFunction.apply(eng, batches[i % 4], phi, ves[i])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"two streams: {1e3 * (t1 - t0) / n:.3f} ms/batch")
