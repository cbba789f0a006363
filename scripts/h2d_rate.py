#!/usr/bin/env python3
"""Development aid: host-to-device copy rate from page-locked memory, one stream or several at once, idle GPU or under a kernel that
fills every compute unit (what the device chunk loader's uploads compete with).   python scripts/h2d_rate.py"""
import time

import torch

dev = torch.device("cuda:0")
mb = 64
host = [torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
dst = [torch.empty(mb << 20, dtype=torch.uint8, device=dev) for _ in range(4)]
a = torch.randn(8192, 8192, device=dev)


def run(nstreams, busy, reps=8):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if busy:
        for _ in range(12):
            torch.mm(a, a)  # ~90 ms of matrix work on the default stream
    for r in range(reps):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                dst[i].copy_(host[i], non_blocking=True)
    for s in streams:
        s.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return reps * nstreams * mb / 1024 / (t1 - t0)


for busy in (False, True):
    for n in (1, 2, 4):
        run(n, busy, 2)
        print(f"{'busy' if busy else 'idle'} GPU, {n} stream(s): {run(n, busy):.1f} GB/s", flush=True)
