"""development: per-parameter gradient error of a dropout training step against the oracle given the masks"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import artifact_oracle as O
from tests.helpers import config_for
from tests import test_dropout_gpu as T
from permutect_amd.data.batch import Batch

family = sys.argv[1] if len(sys.argv) > 1 else "p0"
model = T.dropout_model(family)
ints, floats, packed = T.small_batch(21)
batch = Batch.from_arrays(ints, floats, packed).copy_to(torch.device("cuda"))
out, losses, grads, seed = T.train_step(model, batch)
cfg = config_for(family + "_dropout")
cfg.num_sources = model.num_sources
cfg.dropout = T.mask_provider(model, seed)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
ref_out, ref_losses, ref_grads = O.train_step_grads(sd, cfg, T.oracle_batch(ints, floats, packed))
for n, g in grads.items():
    r = ref_grads[n].numpy()
    e = np.abs(g - r).max()
    print(f"{n:70s} {np.abs(r).max():10.3e} {e:10.3e} {'BAD' if e > 1e-3 * max(np.abs(r).max(), 1e-6) else ''}")
