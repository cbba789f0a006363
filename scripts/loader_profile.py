#!/usr/bin/env python3
"""Where the device chunk loader spends its time (development aid): chunk staging / upload and per-batch composition."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.memory_mapped_data import MemoryMappedData  # noqa: E402
from permutect_amd.data.reads_dataset import ChunkBatch, DeviceChunk, ReadsDataset  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    n, chunk_n, b = 1 << 19, 1 << 18, 65536
    ints, floats, packed = synth_arrays(rng, n, "wgs")
    ds = ReadsDataset(MemoryMappedData.from_arrays(ints, floats, packed))
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    model.train(False)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        chunk = DeviceChunk(ds, 0, chunk_n, dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"chunk of {chunk_n} variants, {chunk.nbytes / 1e6:.0f} MB: {1e3 * (t1 - t0):.1f} ms "
              f"({chunk.nbytes / (t1 - t0) / 1e9:.1f} GB/s)", flush=True)
    ids_all = rng.permutation(chunk_n)
    for rep in range(4):
        ids = ids_all[rep * b:(rep + 1) * b]
        ts = [time.perf_counter()]
        cb = ChunkBatch(chunk, ids); torch.cuda.synchronize(); ts.append(time.perf_counter())
        cb.read_index(); torch.cuda.synchronize(); ts.append(time.perf_counter())
        cb.plan(allow_split=True); ts.append(time.perf_counter())
        cb.plan(allow_split=True).on(dev); torch.cuda.synchronize(); ts.append(time.perf_counter())
        with torch.inference_mode():
            model.compute_batch_output(cb)
        torch.cuda.synchronize(); ts.append(time.perf_counter())
        d = [1e3 * (ts[i + 1] - ts[i]) for i in range(len(ts) - 1)]
        print("batch: compose %.2f ms, read_index %.2f, plan(host) %.2f, plan upload %.2f, forward %.2f" % tuple(d), flush=True)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def host_profile():
    """Host time per filter step with the loader (no synchronisation inside the loop) and where it goes (cProfile)."""
    import cProfile
    import pstats
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    ints, floats, packed = synth_arrays(rng, 1 << 20, "wgs")
    ds = ReadsDataset(MemoryMappedData.from_arrays(ints, floats, packed))
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    model.train(False)

    def run(nsteps):
        it = iter(ds.device_loader(65536, dev, chunk_variants=1 << 18, rng=rng))
        t0 = time.perf_counter()
        for _ in range(nsteps):
            cb = next(it)
            with torch.inference_mode():
                model.compute_batch_output(cb)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return 1e3 * (t1 - t0) / nsteps, 1e3 * (t2 - t0) / nsteps

    run(8)
    print("host enqueue %.2f ms/step, with final sync %.2f ms/step" % run(12), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    run(12)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "host":
    host_profile()
