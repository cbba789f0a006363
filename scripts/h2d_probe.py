import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from permutect_amd.engine import lib as L
lib = L.load()
dev = torch.device("cuda:0")
n = 100 << 20
src = np.random.randint(0, 255, n, dtype=np.uint8)
for rep in range(3):
    t0 = time.perf_counter(); host = torch.empty(n, dtype=torch.uint8, pin_memory=True); t1 = time.perf_counter()
    lib.pmt_host_copy(host.data_ptr(), src.ctypes.data, n, 6); t2 = time.perf_counter()
    d = host.to(dev, non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"pinned alloc {1e3*(t1-t0):.2f} ms, stage {n/(t2-t1)/1e9:.1f} GB/s, H2D {n/(t3-t2)/1e9:.1f} GB/s", flush=True)
    del host
t0 = time.perf_counter(); d2 = torch.from_numpy(src).to(dev); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"pageable H2D {n/(t1-t0)/1e9:.1f} GB/s")
import os
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
