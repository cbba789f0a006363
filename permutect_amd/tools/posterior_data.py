"""The artifact model's hand-off to the posterior model in `filter_variants` (reference tools/filter_variants.py:292-320,
:350-357): every candidate becomes a Datum WITHOUT reads whose info array is its embedding and whose
CACHED_ARTIFACT_LOGIT is the model's logit.

The reference does this one variant at a time in Python (`logits_b.tolist()`, `features_be.cpu()`, a Datum per variant,
amortised-growth memory maps): at 5 M candidates that loop dwarfs the GPU forward.  Here a batch's rows are transformed
as arrays on the device and appended to preallocated host arrays with one copy per batch."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from permutect_amd.data.datum import Data, INFO_START_IDX
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.data.reads_dataset import ReadsDataset


def posterior_rows(int_tensor: torch.Tensor, float_tensor: torch.Tensor, logits_b: torch.Tensor, features_be: torch.Tensor):
    """(int16 [B, 16 + H], float32 [B, 6 + E]) exactly as the reference's per-Datum code leaves them:
    counts zeroed; the logit stored through the float16 scalar array (`Datum.set`, reference data/datum.py:199-211), then
    `set_info_1d(embedding)` replaces the info columns by the float32 embedding, which promotes the row to float32."""
    ints = int_tensor.to(torch.int16).clone()
    ints[:, Data.REF_COUNT.idx] = 0
    ints[:, Data.ALT_COUNT.idx] = 0
    scalars = float_tensor[:, :INFO_START_IDX].to(torch.float16)
    scalars[:, Data.CACHED_ARTIFACT_LOGIT.idx] = logits_b.to(torch.float16)
    floats = torch.cat((scalars.to(torch.float32), features_be.to(torch.float32)), dim=1)
    return ints, floats


@torch.inference_mode()
def make_posterior_mmap(dataset: ReadsDataset, model, batch_size: int, device: Optional[torch.device] = None,
                        chunk_variants: Optional[int] = None) -> MemoryMappedData:
    """`MemoryMappedData.from_generator(generate_posterior_data(...))` of the reference, in dataset order."""
    device = model._device if device is None else torch.device(device)
    n = len(dataset)
    e = model.reducer.output_dimension()
    ints_out = np.zeros((n, dataset._ints.shape[-1]), dtype=np.int16)
    floats_out = np.zeros((n, INFO_START_IDX + e), dtype=np.float32)
    model.train(False)
    done = 0
    for batch in dataset.device_loader(batch_size, device, chunk_variants=chunk_variants, shuffle=False):
        out = model.compute_batch_output(batch)
        ints, floats = posterior_rows(batch.int_tensor, batch.float_tensor, out.logits_b, out.features_be)
        b = batch.size()
        # the loader packs the variants of a batch in the order that fills the workgroups (also without shuffling):
        # rows go back to their place in the dataset
        ints_out[batch.dataset_index] = ints.cpu().numpy()
        floats_out[batch.dataset_index] = floats.cpu().numpy()
        done += b
    assert done == n
    result = MemoryMappedData(ints_out, floats_out, n, None, 0)
    return result
