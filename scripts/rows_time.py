#!/usr/bin/env python3
"""Development aid: device time of the per-variant row MLPs (pmt_rows_forward / pmt_rows_backward) at the bench's batch size."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.engine import lib as L  # noqa: E402
from permutect_amd.engine.runtime import RowsMlpFunction  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
eng = model.engine()
phi = eng.plan.materialize_phi(model)
eng.pack(phi.detach().contiguous())
rng = np.random.default_rng(0)
x_info = torch.from_numpy(rng.standard_normal((n, 71)).astype(np.float32)).to(dev)
x_feat = torch.from_numpy(rng.standard_normal((n, 10)).astype(np.float32)).to(dev).requires_grad_(True)


def timed(fn, k=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3


def info_fwd():
    with torch.inference_mode():
        RowsMlpFunction.apply(eng, L.ROWS_INFO, x_info, eng.trigger, None)


def info_train():
    y = RowsMlpFunction.apply(eng, L.ROWS_INFO, x_info, eng.trigger, None)
    y.sum().backward()


def alt_train():
    y = RowsMlpFunction.apply(eng, L.ROWS_ALT_COUNT, x_feat, eng.trigger, 0.01)
    y.sum().backward()


print(f"n = {n}: info forward (inference) {timed(info_fwd):.1f} us; info forward + backward {timed(info_train):.1f} us; "
      f"alt-count forward + backward (+ d input) {timed(alt_train):.1f} us")
