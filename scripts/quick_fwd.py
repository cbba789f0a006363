"""Quick forward timing (development aid): python scripts/quick_fwd.py [B]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
nref, nalt = rng.integers(0, 11, B), rng.integers(1, 16, B)
ints = np.zeros((B, 58), dtype=np.int16); ints[:, 0], ints[:, 1] = nref, nalt; ints[:, 16:] = rng.integers(0, 5, (B, 42))
floats = np.zeros((B, 77), dtype=np.float16); floats[:, 6:] = rng.standard_normal((B, 71)).astype(np.float16)
R = int(nref.sum() + nalt.sum())
packed = rng.integers(0, 256, (R, 12), dtype=np.uint8)
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
print("B", B, "R", R, "groups", batch.plan().num_groups, "tiles", batch.plan().total_tiles, "fill", R / (16 * batch.plan().total_tiles))
with torch.no_grad():
    for _ in range(3): out = model.compute_batch_output(batch)
    torch.cuda.synchronize()
    eng = model.engine()
    ve = model.variant_embedding(batch); phi = eng.plan.materialize_phi(model)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): out = model.compute_batch_output(batch)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 10
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): eng.forward(batch, phi, ve, False)
    e.record(); torch.cuda.synchronize()
    t_k = s.elapsed_time(e) / 10 / 1e3
flops = 66.3e3 * R
print(f"end-to-end {t_all*1e3:.3f} ms/step -> {B/t_all/1e6:.2f} M read-sets/s ; kernel(+pack) {t_k*1e3:.3f} ms -> {B/t_k/1e6:.2f} M read-sets/s, {flops/t_k/1e12:.2f} TFLOP/s algorithmic ({flops/t_k/157.3e12*100:.1f}% of fp32 MFMA peak)")
