"""Event timeline of ONE workgroup of pmt_forward_kernel (development aid; build with `make -C permutect_amd/csrc
EXTRA=-DPMT_FWD_TRACE=1` after touching pmt_forward.hip).  `python scripts/fwd_trace.py [workgroup] [train|filter]`"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params

wg = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
train = (sys.argv[2] if len(sys.argv) > 2 else "filter") == "train"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(train)
batch = Batch.from_arrays(*synth_arrays(np.random.default_rng(0), 65536, "wgs"), pack=True).copy_to(dev)
eng = model.engine()
NAMES = {1: "start", 2: "setup (offsets, LDS zero, barrier)", 5: "tile meta + decode", 6: "read MLP first linear", 3: "read MLP skip blocks", 4: "concat", 10: "blk: LN, proj1, SELU, LN(h), z2 sums", 11: "blk: barrier", 12: "blk: gate + proj2",
         19: "stash x_L", 21: "reducer + rotation", 22: "set sums + head", 23: "barrier"}
for it in range(3):
    eng.plan.debug_flags.zero_()
    eng.plan.debug_flags[2] = wg + 1 if it == 2 else 0
    with torch.set_grad_enabled(train):
        model.compute_batch_output(batch)
    torch.cuda.synchronize()
log = eng.plan.debug_flags[64:].cpu().numpy().reshape(8, 256, 2)
tot = {}
for w in range(8):
    for i in range(1, 250):
        ev, t = int(log[w, i, 0]), int(np.uint32(log[w, i, 1]))
        if ev == 0: break
        dt = (t - int(np.uint32(log[w, i - 1, 1]))) & 0xFFFFFFFF
        tot[NAMES.get(ev, str(ev))] = tot.get(NAMES.get(ev, str(ev)), 0) + dt / 8
whole = sum(tot.values())
print(f"mean over waves, total {whole:.0f} ticks ({'train' if train else 'filter'})")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {v:9.0f}  {100 * v / whole:5.1f} %  {k}")
for w in (0, 7):
    print("wave", w, [(int(log[w, i, 0]), int((np.uint32(log[w, i, 1]) - np.uint32(log[w, 0, 1])) & 0xFFFFFFFF)) for i in range(0, 30) if log[w, i, 0]])
