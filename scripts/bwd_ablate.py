"""Backward-kernel ablation timing (development aid): runs the train step with parts of pmt_backward_kernel disabled
through PmtBatch.debug_flags[1] (results are wrong when a bit is set; only the kernel time matters).

The production build compiles these switches OUT (pmt_backward.hip: PMT_BWD_DEBUG): build a development library first, e.g.
`make -C permutect_amd/csrc EXTRA="-DPMT_BWD_ABLATE=1 -DPMT_BWD_PROF=1"` into a scratch copy, and load it through PMT_LIB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params

B = 65536
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
ints, floats, packed = synth_arrays(np.random.default_rng(0), B, "wgs")
batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
eng = model.engine()
MASKS = [(0, "full"), (16384, "small params: pushes only, no global atomics"), (16, "no small-param work"), (0, "full"), (16384, "small params: pushes only, no global atomics")] if os.environ.get("PMT_ABLATE_AUX") else [] if os.environ.get("PMT_PROFILE_ONLY") else [(0, "full"), (2048, "no proj1 recompute"), (4096, "no skip-block s1 recompute"), (6144, "neither"), (0, "full")] if os.environ.get("PMT_RECOMPUTE") else None
for mask, name in MASKS if MASKS is not None else [(0, "full"), (64, "no stash prefetch"), (256, "stash reads from L2 (wrong results)"), (256 + 64, "same, no prefetch"), (0, "full again"), (2, "no flush"), (1, "no wgrad"), (4, "no blocks"), (5, "no blocks, no wgrad"), (16, "no small-param atomics")] + ([(512, "exchange: no MFMAs"), (1024, "exchange: no staging"), (1536, "exchange: neither"), (1536 + 2, "exchange: neither, no flush"), (16384, "small params: pushes only, no global atomics"), (0, "full")] if os.environ.get("PMT_ABLATE_EXCHANGE") else []):
    eng.plan.debug_flags[1] = mask
    ts = []
    for i in range(6):
        out = model.compute_batch_output(batch)
        loss = model.compute_batch_losses(out, batch).total_loss
        eng.timers = {"pmt_forward": [], "pmt_backward": []}
        loss.backward()
        torch.cuda.synchronize()
        s, e = eng.timers["pmt_backward"][0]
        ts.append(s.elapsed_time(e))
        eng.timers = None
        eng.space.gtheta.zero_()
    print(f"{name:22s} backward kernel {np.median(ts[2:]):.2f} ms")
# cycle profile of the full kernel (dbg bit 3)
eng.plan.debug_flags.zero_()
eng.plan.debug_flags[1] = 8
out = model.compute_batch_output(batch)
model.compute_batch_losses(out, batch).total_loss.backward()
torch.cuda.synchronize()
prof = eng.plan.debug_flags[8:56].cpu().numpy().view(np.uint64)
tot = float(prof[7])
for i, nm in enumerate(["mlp wgrad accumulate", "mlp wgrad barrier", "mlp wgrad flush", "-", "tail recompute + head", "rotation", "reducer backward", "whole kernel (sum over waves)", "blk p1 recompute proj1", "blk p2 dgrad proj2 + gate", "blk proj2 wgrad", "blk set coupling", "blk p3 LN(h)/selu bwd", "blk proj1 wgrad", "blk dgrad proj1 + LN(D) bwd", "-", "split + read MLP backward", "wgrad exchange: barrier before staging", "wgrad exchange: split + stage stores", "wgrad exchange: barrier after staging"]):
    if prof[i]: print(f"  {nm:32s} {prof[i]:.3e} cycles  {100 * prof[i] / tot:5.1f} %")
