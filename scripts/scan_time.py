#!/usr/bin/env python3
"""pmt_scan_counts timing on a batch-shaped int64 table (development aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permutect_amd.engine import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
for n in (8192, 65536, 1 << 20):
    table = torch.randint(0, 12, (n, 58), dtype=torch.int64, device=dev)
    ro = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ao = torch.empty(n + 1, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    ts = []
    for i in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.pmt_scan_counts(table[:, 0].data_ptr(), table[:, 1].data_ptr(), 8, table.stride(0), n, ro.data_ptr(), ao.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ok = torch.equal(ro[1:].long(), torch.cumsum(table[:, 0], 0)) and torch.equal(ao[1:].long(), torch.cumsum(table[:, 1], 0))
    print(f"n = {n}: {sorted(ts)[len(ts) // 2]:.1f} us, correct = {ok}")
