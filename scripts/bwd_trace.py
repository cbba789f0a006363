"""Event timeline of ONE workgroup of pmt_backward_kernel (development aid): every wave logs (event, clock) at the phase
boundaries (trace_ev in pmt_bwd_device.hpp / pmt_backward.hip; build with `make -C permutect_amd/csrc EXTRA=-DPMT_BWD_TRACE=1`
after touching pmt_backward.hip); prints, per wave, the time between consecutive events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params

B = 65536
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
batch = Batch.from_arrays(*synth_arrays(np.random.default_rng(0), B, "wgs"), pack=True).copy_to(dev)
eng = model.engine()
NAMES = {1: "start", 14: "tail+head done", 15: "rotation done", 16: "reducer bwd done", 17: "END", 18: "blk p1 recompute proj1", 19: "blk p2 dgrad proj2+gate",
         20: "blk proj2 wgrad", 21: "blk set coupling", 22: "blk p3 LN(h)/selu bwd", 23: "blk proj1 wgrad", 24: "blk dgrad proj1+LN bwd", 26: "read MLP bwd done",
         100: "x: at barrier 1", 101: "x: past barrier 1", 102: "x: staged", 103: "x: past barrier 2", 104: "x: contracted", 105: "x: emitted",
         220: "skip: s1 ready", 221: "skip: d1 ready", 222: "skip: s0 ready"}
for it in range(3):
    eng.plan.debug_flags.zero_()
    eng.plan.debug_flags[2] = wg + 1 if it == 2 else 0
    out = model.compute_batch_output(batch)
    model.compute_batch_losses(out, batch).total_loss.backward()
    torch.cuda.synchronize()
log = eng.plan.debug_flags[64:].cpu().numpy().reshape(8, 256, 2)
t0 = min(int(np.uint32(log[w, 0, 1])) for w in range(8))
for w in (0, 4, 7):
    print(f"--- wave {w}")
    prev = None
    for i in range(250):
        ev, t = int(log[w, i, 0]), int(np.uint32(log[w, i, 1]))
        if ev == 0: break
        rel = (t - t0) & 0xFFFFFFFF
        name = NAMES.get(ev, f"mlp op {ev - 200}: load input" if 200 <= ev < 220 else str(ev))
        print(f"  {rel:9d}  +{(rel - prev) if prev is not None else 0:7d}  {name}")
        prev = rel
# totals per kind of segment, averaged over the waves
tot = {}
for w in range(8):
    for i in range(1, 250):
        ev, t = int(log[w, i, 0]), int(np.uint32(log[w, i, 1]))
        if ev == 0: break
        dt = (t - int(np.uint32(log[w, i - 1, 1]))) & 0xFFFFFFFF
        key = f"{NAMES.get(int(log[w, i - 1, 0]), 'mlp load' if 200 <= log[w, i - 1, 0] < 220 else str(log[w, i - 1, 0]))} -> {NAMES.get(ev, 'mlp load' if 200 <= ev < 220 else str(ev))}"
        tot[key] = tot.get(key, 0) + dt / 8
whole = sum(tot.values())
print(f"--- mean over waves, total {whole:.0f} ticks")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {v:9.0f}  {100 * v / whole:5.1f} %  {k}")
