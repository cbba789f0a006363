#!/usr/bin/env python3
"""Development aid: the device timeline of the filter forward fed by the device chunk loader: how busy the compute queue is, and
what sits on either side of its idle gaps."""
import os
import sys

import numpy as np
import torch
from torch.autograd import DeviceType
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.memory_mapped_data import MemoryMappedData  # noqa: E402
from permutect_amd.data.reads_dataset import ReadsDataset  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402

dev = torch.device("cuda:0")
ints, floats, packed = synth_arrays(np.random.default_rng(0), 1 << 21, "wgs")
ds = ReadsDataset(MemoryMappedData.from_arrays(ints, floats, packed)).pin_memory()
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.engine()
model.train(False)


def run():
    with torch.inference_mode():
        for cb in ds.device_loader(65536, dev, chunk_variants=1 << 18, shuffle=False):
            model.compute_batch_output(cb)
    torch.cuda.synchronize()


run()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    run()
ev = [(e.time_range.start, e.time_range.end, e.name) for e in prof.events() if e.device_type == DeviceType.CUDA]
ev.sort()
kern = [e for e in ev if "Memcpy" not in e[2] and "Memset" not in e[2]]
cpy = [e for e in ev if "Memcpy" in e[2]]
t0, t1 = kern[0][0], max(e[1] for e in kern)
busy, cur_s, cur_e = 0.0, kern[0][0], kern[0][1]
gaps = []
last_name = kern[0][2]
for s, e, n in kern[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n
busy += cur_e - cur_s
print(f"span {1e-3 * (t1 - t0):.2f} ms for 32 batches = {1e-3 * (t1 - t0) / 32:.3f} ms/batch; compute queue busy {100 * busy / (t1 - t0):.1f} %; "
      f"H2D copies {1e-3 * sum(e[1] - e[0] for e in cpy):.2f} ms in {len(cpy)} pieces")
print("largest idle gaps on the compute queue (us, after -> before, at ms):")
for g, a, b, at in sorted(gaps, reverse=True)[:14]:
    print(f"  {g:8.1f}  {a[:50]:50s} -> {b[:50]:50s} @ {1e-3 * at:7.2f}")
tot = {}
for g, a, b, at in gaps:
    key = (a[:40], b[:40])
    tot[key] = tot.get(key, 0.0) + g
print("idle time by (after, before), ms:")
for (a, b), g in sorted(tot.items(), key=lambda kv: -kv[1])[:10]:
    print(f"  {1e-3 * g:7.2f}  {a} -> {b}")
