"""Batch / DownsampledBatch: the input contract of the hot path (reference permutect/data/batch.py:41-174, 383-459).

Differences that matter on MI355X, none of which change results:
  * compressed reads stay PACKED (uint8 [R, 7 + nf]) all the way into HBM; the HIP forward kernel decodes them in its
    load stage (12 B/read over PCIe and HBM instead of 244 B/read as fp32).  `get_reads_re()` still returns the
    reference's float16 view for callers that want it.
  * the batch carries a host-side *group plan* (pmt_plan_groups) computed from the counts it already has on the host
    at collate time, so the forward needs no device->host sync (the reference syncs twice per forward,
    artifact_model.py:241 and sets/ragged_sets.py:37).
The two reference quirks at this boundary are reproduced bit-exactly: the uint8 decode wrap
(data/plain_text_data.py:510-511) and DownsampledBatch's un-offset alt gather (data/batch.py:436-439).
"""
from __future__ import annotations

import copy
import os
import ctypes as C
from random import randint
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor

from permutect_amd.data.datum import Data, Datum, HAPLOTYPES_START_IDX, INFO_START_IDX, NUMBER_OF_BYTES_IN_PACKED_READ
from permutect_amd.engine import lib as L
from permutect_amd.enums import Label


def decode_packed_reads(packed: np.ndarray) -> np.ndarray:
    """uint8 [R, 7 + nf] -> float16 [R, 56 + nf] exactly as the reference collate does (including the uint8 wrap)."""
    bits = np.unpackbits(packed[:, :NUMBER_OF_BYTES_IN_PACKED_READ], axis=1).astype(np.float16)
    wrapped = (packed[:, NUMBER_OF_BYTES_IN_PACKED_READ:] - np.uint8(128)).astype(np.float32)  # uint8 arithmetic wraps
    return np.hstack((bits, (wrapped / 32.0).astype(np.float16)))


def pack_order(ref_counts: np.ndarray, alt_counts: np.ndarray, window: int = 64) -> np.ndarray:
    """An order of a batch's variants that packs fuller workgroups (pmt_pack_order): a workgroup costs the same full or not,
    and the order of the variants inside a batch means nothing to the model (the reference shuffles them)."""
    rc = np.ascontiguousarray(ref_counts, dtype=np.int32)
    ac = np.ascontiguousarray(alt_counts, dtype=np.int32)
    order = np.empty(len(rc), dtype=np.int32)
    L.check(L.load().pmt_pack_order(rc.ctypes.data, ac.ctypes.data, len(rc), window, order.ctypes.data), "pmt_pack_order")
    return order.astype(np.int64)


def _segment_rows(starts: np.ndarray, lengths: np.ndarray) -> np.ndarray:
    """Concatenation of the row ranges [starts[i], starts[i] + lengths[i])."""
    total = int(lengths.sum())
    out_start = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(lengths, out=out_start[1:])
    return np.repeat(starts - out_start[:-1], lengths) + np.arange(total, dtype=np.int64)


class GroupPlan:
    """Partition of the batch's variants into register-resident groups (host arrays + device copies)."""

    def __init__(self, ref_counts: np.ndarray, alt_counts: np.ndarray, allow_split: bool = False):
        lib = L.load()
        b = len(ref_counts)
        rc = np.ascontiguousarray(ref_counts, dtype=np.int32)
        ac = np.ascontiguousarray(alt_counts, dtype=np.int32)
        gs = np.zeros(b + 1, dtype=np.int32)
        gt = np.zeros(b + 1, dtype=np.int32)
        bad = C.c_int32(-1)
        n = lib.pmt_plan_groups(rc.ctypes.data, ac.ctypes.data, b, gs.ctypes.data, gt.ctypes.data, C.byref(bad))
        self.span = None  # [G, 6] when some read set is split over several groups ("layered" execution)
        if n == L.E_CAPACITY:
            if not allow_split:
                raise L.PmtError(
                    f"variant {bad.value} has {int(rc[bad.value])} ref / {int(ac[bad.value])} alt reads: more than the "
                    f"{L.GROUP_TILES * L.TILE}-read register-resident group capacity of the gfx950 kernels")
            max_groups = b + int((rc.astype(np.int64).sum() + ac.astype(np.int64).sum()) // (L.TILE * L.GROUP_TILES // 2)) + 8
            span = np.zeros((max_groups, 6), dtype=np.int32)
            gt = np.zeros(max_groups + 1, dtype=np.int32)
            layered = C.c_int32(0)
            n = lib.pmt_plan_groups_split(rc.ctypes.data, ac.ctypes.data, b, span.ctypes.data, gt.ctypes.data, max_groups, C.byref(layered))
            L.check(n, "pmt_plan_groups_split")
            self.span = span[:n].copy()
            gs = np.zeros(n + 1, dtype=np.int32)  # unused by the layered kernels
            # how many groups cover each variant (PmtBatch.set_groups): what the groups of a split read set wait for when they join
            # their per-set sums inside ONE launch
            cover = np.zeros(b + 1, dtype=np.int64)
            np.add.at(cover, self.span[:, 0], 1)
            np.add.at(cover, self.span[:, 1], -1)
            self.set_groups = np.cumsum(cover)[:b].astype(np.int32)
        L.check(n, "pmt_plan_groups")
        self.num_groups = n
        self.group_start = gs[: n + 1].copy()
        self.group_tile_base = gt[: n + 1].copy()
        self.total_tiles = int(gt[n])
        self._dev = {}

    @classmethod
    def from_prepared(cls, group_start: np.ndarray, group_tile_base: np.ndarray, device_views=None) -> "GroupPlan":
        """A plan that pmt_prepare_chunk computed (whole read sets per group, no split): host arrays of g + 1 ints each and,
        optionally, their copies already on the device as {str(device): (group_start, group_tile_base, None)}."""
        self = cls.__new__(cls)
        self.span = None
        self.num_groups = len(group_start) - 1
        self.group_start, self.group_tile_base = group_start, group_tile_base
        self.total_tiles = int(group_tile_base[-1])
        self._dev = dict(device_views or {})
        return self

    @property
    def layered(self) -> bool:
        return self.span is not None

    set_groups = None  # int32 [B] for a split plan (span is not None)

    def use_span(self, span: np.ndarray, group_tile_base: np.ndarray, num_variants: int):
        """Replace the plan by explicit group spans ([G, 6]: v0, v1, ref_begin, ref_end, alt_begin, alt_end) -- tests cut read
        sets into groups of their own choosing -- keeping everything derived from them consistent."""
        self.span = np.ascontiguousarray(span, dtype=np.int32)
        self.num_groups = len(self.span)
        self.group_tile_base = np.ascontiguousarray(group_tile_base, dtype=np.int32)
        self.total_tiles = int(self.group_tile_base[-1])
        self.group_start = np.zeros(self.num_groups + 1, dtype=np.int32)
        cover = np.zeros(num_variants + 1, dtype=np.int64)
        np.add.at(cover, self.span[:, 0], 1)
        np.add.at(cover, self.span[:, 1], -1)
        self.set_groups = np.cumsum(cover)[:num_variants].astype(np.int32)
        self._dev = {}

    def set_groups_on(self, device: torch.device):
        if self.set_groups is None:
            return None
        key = "sets:" + str(device)
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(self.set_groups).to(device)
        return self._dev[key]

    def on(self, device: torch.device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self.group_start).to(device), torch.from_numpy(self.group_tile_base).to(device),
                              None if self.span is None else torch.from_numpy(self.span).to(device))
        return self._dev[key]


class DevicePlan:
    """A DownsampledBatch's own plan, made ON THE DEVICE from the counts it kept (pmt_plan_groups_device: they are decided there and
    never come to the host).  `num_groups` / `total_tiles` are CAPACITIES -- the launch grid, the stash size -- the kernels read the
    real group count from `num_groups_dev` (PmtBatch.num_groups_dev).  Capacity: next-fit packing of smaller counts in the same order
    never makes more groups than the parent's plan, plus one per chunk boundary of the device planner; tiles: every read in a tile of
    its own side and group, i.e. reads / 16 + 2 per group at most -- bounded by the parent's tiles + 2 per group of capacity."""
    layered, span, set_groups = False, None, None

    def __init__(self, parent_plan: GroupPlan, ref_offsets: Tensor, alt_offsets: Tensor, num_variants: int, fault: Optional[Tensor]):
        lib, dev = L.load(), ref_offsets.device
        self.num_groups = parent_plan.num_groups + int(lib.pmt_plan_device_chunks(num_variants))
        self.total_tiles = parent_plan.total_tiles + 2 * self.num_groups
        self._gs = torch.empty(self.num_groups + 1, dtype=torch.int32, device=dev)
        self._gt = torch.empty(self.num_groups + 1, dtype=torch.int32, device=dev)
        self.num_groups_dev = torch.empty(1, dtype=torch.int32, device=dev)
        scratch = torch.empty(2 * int(lib.pmt_plan_device_chunks(num_variants)), dtype=torch.int32, device=dev)
        L.check(lib.pmt_plan_groups_device(ref_offsets.data_ptr(), alt_offsets.data_ptr(), num_variants, self._gs.data_ptr(), self._gt.data_ptr(),
                                           self.num_groups, self.num_groups_dev.data_ptr(), None if fault is None else fault.data_ptr(),
                                           scratch.data_ptr(), L.raw_stream(dev)), "pmt_plan_groups_device")
        self._keep = (ref_offsets, alt_offsets, fault, scratch)

    def on(self, device: torch.device):
        assert self._gs.device == torch.device(device) or str(self._gs.device) == str(device)
        return self._gs, self._gt, None

    def set_groups_on(self, device: torch.device):
        return None


class Batch:
    order = None  # set by from_arrays(pack=True): batch position -> row of the arrays the batch was built from

    def __init__(self, data: List[Datum]):
        ints = np.vstack([d.get_int_array() for d in data])
        floats = np.vstack([d.get_float_array() for d in data])
        reads = np.vstack([d.get_ref_reads_re() for d in data] + [d.get_alt_reads_re() for d in data])
        self._init_from_arrays(ints, floats, reads)

    @classmethod
    def from_arrays(cls, int_array: np.ndarray, float_array: np.ndarray, reads: np.ndarray, pack: bool = False) -> "Batch":
        """int_array [B, 16+H] (int16), float_array [B, 6+I] (float16), reads in batch order (all ref rows of all
        variants, then all alt rows): uint8 [R, 7+nf] packed, or float16 [R, F].  `pack`: put the variants in the order
        that fills the kernels' workgroups best (`pack_order`); `self.order` then maps batch positions to rows of the
        arrays given."""
        self = cls.__new__(cls)
        self.order = None
        if pack and len(int_array) > 1:
            ref = np.asarray(int_array[:, Data.REF_COUNT.idx]).astype(np.int64)
            alt = np.asarray(int_array[:, Data.ALT_COUNT.idx]).astype(np.int64)
            order = pack_order(ref, alt)
            ref_start = np.concatenate([[0], np.cumsum(ref)[:-1]])
            alt_start = int(ref.sum()) + np.concatenate([[0], np.cumsum(alt)[:-1]])
            rows = np.concatenate([_segment_rows(ref_start[order], ref[order]), _segment_rows(alt_start[order], alt[order])])
            int_array, float_array, reads = int_array[order], float_array[order], reads[rows]
            self.order = order
        self._init_from_arrays(int_array, float_array, reads)
        return self

    def _init_from_arrays(self, ints: np.ndarray, floats: np.ndarray, reads: np.ndarray):
        self.int_tensor = torch.from_numpy(np.ascontiguousarray(ints)).to(torch.long)
        self.float_tensor = torch.from_numpy(np.ascontiguousarray(floats)).to(torch.float)
        if reads.dtype == np.uint8:
            self.packed_reads: Optional[Tensor] = torch.from_numpy(np.ascontiguousarray(reads))
            self.reads_re: Optional[Tensor] = None
            self._num_read_features = 8 * NUMBER_OF_BYTES_IN_PACKED_READ + reads.shape[1] - NUMBER_OF_BYTES_IN_PACKED_READ
        else:
            self.packed_reads = None
            self.reads_re = torch.from_numpy(np.ascontiguousarray(reads))
            self._num_read_features = reads.shape[1]
        self._size = len(self.int_tensor)
        ref = self.int_tensor[:, Data.REF_COUNT.idx].numpy()
        alt = self.int_tensor[:, Data.ALT_COUNT.idx].numpy()
        assert int(ref.sum() + alt.sum()) == reads.shape[0], "read rows do not match the counts"
        self._host_counts = (ref.astype(np.int32), alt.astype(np.int32))
        self._plan: Optional[GroupPlan] = None
        self._offsets = None

    # ---- reference accessors ---------------------------------------------------------------------------------------
    def get(self, field: Data) -> Tensor:
        return (self.int_tensor if field.kind == "int" else self.float_tensor)[:, field.idx]

    def get_training_labels(self) -> Tensor:
        labels = self.get(Data.LABEL)
        return 1.0 * (labels == Label.ARTIFACT) + 0.5 * (labels == Label.UNLABELED)

    def get_is_labeled_mask(self) -> Tensor:
        return (self.get(Data.LABEL) != Label.UNLABELED).int()

    def get_info_be(self) -> Tensor:
        return self.float_tensor[:, INFO_START_IDX:]

    def get_haplotypes_bs(self) -> Tensor:
        return self.int_tensor[:, HAPLOTYPES_START_IDX:]

    def get_one_hot_haplotypes_bcs(self) -> Tensor:
        hap = self.get_haplotypes_bs()
        b, h = hap.shape
        one_hot = torch.nn.functional.one_hot(hap, num_classes=5)  # [B, H, 5]
        return one_hot.permute(0, 2, 1).reshape(b, 10, h // 2)  # refA, altA, refC, altC, ...

    def get_reads_re(self) -> Tensor:
        """float view of the reads, [R, F] (decoded on demand when the batch holds packed bytes)."""
        if self.reads_re is not None:
            return self.reads_re
        dec = decode_packed_reads(self.packed_reads.cpu().numpy())
        return torch.from_numpy(dec).to(self.packed_reads.device)

    def num_read_features(self) -> int:
        return self._num_read_features

    def size(self) -> int:
        return self._size

    def pin_memory(self):
        self.int_tensor = self.int_tensor.pin_memory()
        self.float_tensor = self.float_tensor.pin_memory()
        if self.packed_reads is not None:
            self.packed_reads = self.packed_reads.pin_memory()
        if self.reads_re is not None:
            self.reads_re = self.reads_re.pin_memory()
        return self

    def copy_to(self, device, dtype=torch.float32) -> "Batch":
        nb = device.type == "cuda"
        new = copy.copy(self)
        new.int_tensor = self.int_tensor.to(device, non_blocking=nb)
        new.float_tensor = self.float_tensor.to(device, non_blocking=nb)
        if self.packed_reads is not None:
            new.packed_reads = self.packed_reads.to(device, non_blocking=nb)
        if self.reads_re is not None:
            new.reads_re = self.reads_re.to(device=device, dtype=dtype, non_blocking=nb)
        new._offsets = None
        return new

    # ---- engine-side views -----------------------------------------------------------------------------------------
    def host_counts(self):
        if self._host_counts is None:  # a batch assembled on the device: one sync, like the reference's .item()
            ints = self.int_tensor[:, :2].cpu().numpy()
            self._host_counts = (ints[:, 0].astype(np.int32), ints[:, 1].astype(np.int32))
        return self._host_counts

    def plan(self, allow_split: bool = False, fault: Optional[Tensor] = None) -> GroupPlan:
        """`allow_split`: read sets beyond one workgroup are split over several groups instead of refused
        (pmt_forward_layered / pmt_backward_layered run such a plan).  `fault`: see DownsampledBatch.plan."""
        if self._plan is None:
            self._plan = GroupPlan(*self.host_counts(), allow_split=allow_split)
        return self._plan

    def device_counts(self):
        """(ref_counts, alt_counts, elem_bytes, stride_elems) as device tensors for pmt_scan_counts."""
        return self.int_tensor[:, Data.REF_COUNT.idx], self.int_tensor[:, Data.ALT_COUNT.idx], 8, self.int_tensor.stride(0)

    def read_rows(self):
        """(tensor, format, row_bytes, gather_index or None)"""
        if self.packed_reads is not None:
            return self.packed_reads, L.READS_PACKED_U8, self.packed_reads.shape[1], None
        r = self.reads_re
        if r.dtype == torch.float16:
            return r, L.READS_F16, 2 * r.shape[1], None
        if r.dtype == torch.float32:
            return r, L.READS_F32, 4 * r.shape[1], None
        raise L.PmtError(f"unsupported reads dtype {r.dtype}")


class DownsampledBatch(Batch):
    """Bernoulli read subsampling of a parent batch without copying reads (reference data/batch.py:383-459).

    The kept-alt indices index the alt-only mask but are used un-offset into the full read array, exactly like the
    reference; `fix_alt_gather=True` opts into the offset (intended) behaviour."""

    def __init__(self, original_batch: Batch, ref_fracs_b: Tensor, alt_fracs_b: Tensor, fix_alt_gather: bool = False):
        self.int_tensor = original_batch.int_tensor
        self.float_tensor = original_batch.float_tensor
        self.device = self.int_tensor.device
        self.packed_reads = original_batch.packed_reads
        self.reads_re = original_batch.reads_re
        self._num_read_features = original_batch._num_read_features
        self._size = original_batch._size
        self._parent = original_batch
        self._offsets = None

        old_ref, old_alt = original_batch.get(Data.REF_COUNT), original_batch.get(Data.ALT_COUNT)
        ref_host, alt_host = original_batch.host_counts()
        total_ref, total_alt = int(ref_host.sum()), int(alt_host.sum())
        ref_probs = torch.repeat_interleave(ref_fracs_b, repeats=old_ref, dim=0, output_size=total_ref)
        alt_probs = torch.repeat_interleave(alt_fracs_b, repeats=old_alt, dim=0, output_size=total_alt)
        keep_ref = torch.zeros(total_ref, device=self.device, dtype=torch.int64)
        keep_ref.bernoulli_(p=ref_probs)
        keep_alt = torch.zeros(total_alt, device=self.device, dtype=torch.int64)
        keep_alt.bernoulli_(p=alt_probs)
        # guarantee one alt read per variant: force one pseudo-randomly chosen alt of each set
        random_int = randint(0, 100)
        alt_ends = torch.cumsum(old_alt, dim=0)
        override = alt_ends - torch.remainder(torch.tensor([random_int], device=self.device, dtype=torch.int64), old_alt) - 1
        keep_alt[override] = 1
        self.ref_counts = torch.segment_reduce(keep_ref.float(), reduce="sum", lengths=old_ref).round().int()
        self.alt_counts = torch.segment_reduce(keep_alt.float(), reduce="sum", lengths=old_alt).round().int()
        kept_ref = torch.nonzero(keep_ref).view(-1)
        kept_alt = torch.nonzero(keep_alt).view(-1)
        if fix_alt_gather:
            kept_alt = kept_alt + total_ref
        self.read_indices = torch.hstack((kept_ref, kept_alt))
        self._host_counts = None

    @classmethod
    def on_device(cls, original_batch: Batch, seed: int, ref_fracs_b: Optional[Tensor] = None,
                  alt_fracs_b: Optional[Tensor] = None, ref_weights_b4: Optional[Tensor] = None,
                  alt_weights_b4: Optional[Tensor] = None, fix_alt_gather: bool = False, force_random: Optional[int] = None,
                  weight_tables=None, num_sources: int = 1):
        """The same sampling scheme in two launches and no host sync (pmt_downsample_*): fractions from the Downsampler's
        Beta mixture (or given), Bernoulli keep per read from a counter-based generator, one forced alt read per variant."""
        self = cls.__new__(cls)
        p = original_batch
        self.int_tensor, self.float_tensor = p.int_tensor, p.float_tensor
        self.device = dev = self.int_tensor.device
        self.packed_reads, self.reads_re = p.packed_reads, p.reads_re
        self._num_read_features, self._size, self._parent = p._num_read_features, p._size, p
        lib = L.load()
        stream = L.raw_stream(dev)
        b = p._size

        def scan(ref_c, alt_c, elem, stride):
            ro = torch.empty(b + 1, dtype=torch.int32, device=dev)
            ao = torch.empty(b + 1, dtype=torch.int32, device=dev)
            L.check(lib.pmt_scan_counts(ref_c.data_ptr(), alt_c.data_ptr(), elem, stride, b, ro.data_ptr(), ao.data_ptr(), stream),
                    "pmt_scan_counts")
            return ro, ao

        if getattr(p, "_offsets", None) is None:
            p._offsets = scan(*p.device_counts())
        pro, pao = p._offsets
        a = L.PmtDownsample()
        a.num_variants, a.reference_alt_gather, a.seed = b, 0 if fix_alt_gather else 1, seed & 0xFFFFFFFFFFFFFFFF
        a.force_random = randint(0, 100) if force_random is None else force_random
        a.ref_offsets, a.alt_offsets = pro.data_ptr(), pao.data_ptr()
        keep = [t if t is None else t.contiguous().float() for t in (ref_weights_b4, alt_weights_b4, ref_fracs_b, alt_fracs_b)]
        a.ref_weights_b4, a.alt_weights_b4, a.ref_fracs_in, a.alt_fracs_in = [None if t is None else t.data_ptr() for t in keep]
        if weight_tables is not None:  # (ref table, alt table) [S 3 V R A][4]: the kernel looks the variant's cell up itself
            from permutect_amd.enums import Variation
            from permutect_amd.training import downsampler as D
            assert ref_weights_b4 is None and alt_weights_b4 is None and ref_fracs_b is None
            a.ref_weight_table, a.alt_weight_table = weight_tables[0].data_ptr(), weight_tables[1].data_ptr()
            cols = [Batch.get(p, f) for f in (Data.LABEL, Data.VARIANT_TYPE, Data.SOURCE)]  # (the PARENT's columns and counts)
            a.labels, a.variant_types, a.sources = [L.int_column(t) for t in cols]
            g = a.bins
            g.num_sources, g.num_variant_types, g.num_ref_bins, g.num_alt_bins = num_sources, len(Variation), D.NUM_REF_COUNT_BINS, D.NUM_ALT_COUNT_BINS
            g.count_bin_skip, g.max_ref_count, g.max_alt_count = D.COUNT_BIN_SKIP, D.MAX_REF_COUNT, D.MAX_ALT_COUNT
            keep = keep + [weight_tables, cols]
        self.ref_fracs = torch.empty(b, dtype=torch.float32, device=dev)
        self.alt_fracs = torch.empty(b, dtype=torch.float32, device=dev)
        self.ref_counts = torch.empty(b, dtype=torch.int32, device=dev)
        self.alt_counts = torch.empty(b, dtype=torch.int32, device=dev)
        L.check(lib.pmt_downsample_counts(C.byref(a), self.ref_fracs.data_ptr(), self.alt_fracs.data_ptr(),
                                          self.ref_counts.data_ptr(), self.alt_counts.data_ptr(), stream), "pmt_downsample_counts")
        self._offsets = scan(self.ref_counts, self.alt_counts, 4, 1)
        ref_host, alt_host = p.host_counts()
        # sized for the parent (an upper bound): the number of kept reads never has to come back to the host
        # (zero-filled: the unused tail must stay a valid row index, it is composed with a parent's gather index)
        self.read_indices = torch.zeros(int(ref_host.sum()) + int(alt_host.sum()), dtype=torch.int64, device=dev)
        L.check(lib.pmt_downsample_index(C.byref(a), self.ref_fracs.data_ptr(), self.alt_fracs.data_ptr(),
                                         self._offsets[0].data_ptr(), self._offsets[1].data_ptr(),
                                         self.read_indices.data_ptr(), stream), "pmt_downsample_index")
        self._host_counts = None
        self._keep = keep
        return self

    def get(self, field: Data) -> Tensor:
        if field == Data.REF_COUNT:
            return self.ref_counts
        if field == Data.ALT_COUNT:
            return self.alt_counts
        return super().get(field)

    def get_reads_re(self) -> Tensor:
        return self._parent.get_reads_re()[self.read_indices]

    def plan(self, allow_split: bool = False, fault: Optional[Tensor] = None):
        """The batch's own plan.  Its counts live on the device; so does its planner (DevicePlan / pmt_plan_groups_device): about half
        the parent's workgroups, hence about half its time -- on the parent's plan (what rounds 1 - 4 did: the parent's counts bound
        these) every group ran half empty at full cost.  `fault`: the engine's fault word, raised by the planner if its capacity
        argument were ever wrong.  PMT_DEVICE_PLAN=0, a CPU batch, or a parent with split read sets: the earlier paths."""
        parent = self._parent.plan(allow_split=allow_split)  # parent counts are upper bounds of the downsampled counts
        if not parent.layered:
            if not self.int_tensor.is_cuda or os.environ.get("PMT_DEVICE_PLAN", "1") == "0" or getattr(self, "_offsets", None) is None or self._size < 1:
                return parent
            if getattr(self, "_device_plan", None) is None:
                self._device_plan = DevicePlan(parent, self._offsets[0], self._offsets[1], self._size, fault)
            return self._device_plan
        if getattr(self, "_own_plan", None) is None:  # split groups name explicit rows: plan from the downsampled counts
            self._own_plan = GroupPlan(*self.host_counts(), allow_split=True)
        return self._own_plan

    def host_counts(self):
        if self._host_counts is None:
            self._host_counts = (self.ref_counts.cpu().numpy().astype(np.int32), self.alt_counts.cpu().numpy().astype(np.int32))
        return self._host_counts

    def device_counts(self):
        return self.ref_counts, self.alt_counts, 4, 1

    def read_rows(self):
        t, fmt, row_bytes, parent_index = self._parent.read_rows()
        # a parent that is itself a gather (a batch composed from a device-resident chunk): compose the two indices -- once per batch
        # (the forward and the backward both ask)
        if parent_index is None:
            return t, fmt, row_bytes, self.read_indices
        if getattr(self, "_composed_index", None) is None:
            self._composed_index = parent_index[self.read_indices]
        return t, fmt, row_bytes, self._composed_index
