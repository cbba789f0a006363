"""CPU: host-side logic -- state_dict contract, checkpoint format, C-ABI surface, group planner, batch collate."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import artifact_oracle as O
from permutect_amd.architecture.artifact_model import ArtifactModel, load_model
from permutect_amd.data.batch import Batch, DownsampledBatch, decode_packed_reads
from permutect_amd.data.datum import Data, Datum
from permutect_amd.engine import lib as L
from permutect_amd.parameters import P0_DIMS, p0_params, t0_params
from tests.helpers import GOLDEN, config_for, load_case

CPU = torch.device("cpu")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,params", [("t0_b8", t0_params), ("p0_b16", p0_params)])
def test_state_dict_matches_reference_keys_and_shapes(name, params):
    z, sd, _ = load_case(name)
    model = ArtifactModel(params(), device=CPU, **P0_DIMS)
    ours = model.state_dict()
    assert list(ours.keys()) == list(sd.keys())  # same names in the same order
    for k in sd:
        assert tuple(ours[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)  # strict
    assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for k, v in sd.items() if not k.endswith(".base"))


def test_dropout_model_keeps_the_reference_state_dict_layout():
    """reference architecture/mlp.py:57-58: with dropout_p > 0 an nn.Dropout sits behind every Linear and shifts the Sequential
    indices in the state_dict keys (p0_dropout_eval.npz: a reference model built with dropout_p = 0.25)."""
    z, sd, _ = load_case("p0_dropout_eval")
    params = p0_params()
    params.dropout_p = float(z["dropout_p"])
    model = ArtifactModel(params, device=CPU, **P0_DIMS)
    assert list(model.state_dict().keys()) == list(sd.keys())
    assert list(sd.keys()) != list(ArtifactModel(p0_params(), device=CPU, **P0_DIMS).state_dict().keys())  # (the indices do shift)
    model.load_state_dict(sd)  # strict


def test_batchnorm_model_keeps_the_reference_state_dict_layout_and_trains_nowhere():
    """reference architecture/mlp.py:52-53: with batch_normalize an nn.BatchNorm1d sits in front of every Linear of the four MLPs
    (p0_batchnorm_eval.npz: a reference model built that way, running statistics away from (0, 1)).  The modules exist so that
    the checkpoint loads key for key; eval mode is folded into the Linears (tests/test_forward_gpu.py runs it), training with
    batch statistics is refused."""
    z, sd, b = load_case("p0_batchnorm_eval")
    params = p0_params()
    params.batch_normalize = True
    model = ArtifactModel(params, device=CPU, **P0_DIMS)
    assert list(model.state_dict().keys()) == list(sd.keys())
    for k in sd:
        assert tuple(model.state_dict()[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)  # strict
    assert sum(isinstance(m, torch.nn.BatchNorm1d) for m in model.modules()) == 20
    model.train(True)
    with pytest.raises(NotImplementedError, match="batch_normalize"):
        model.compute_batch_output(Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]))


def test_oracle_eval_mode_of_a_batchnorm_model_matches_the_reference_outputs():
    z, sd, b = load_case("p0_batchnorm_eval")
    cfg = config_for("p0_batchnorm_eval")
    cfg.batch_normalize = True
    with torch.no_grad():
        out = O.compute_batch_output(sd, cfg, b["reads_re"], b["nref"], b["nalt"], b["info_be"], b["haplotypes_bh"])
    for k in ("logits_b", "features_be", "ref_features_be"):
        ref = z["out/" + k]
        np.testing.assert_allclose(out[k].numpy(), ref, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)


def test_p0_parameter_count():
    model = ArtifactModel(p0_params(), device=CPU, **P0_DIMS)
    assert sum(p.numel() for p in model.parameters()) == 59845


def test_checkpoint_roundtrip_and_dict_keys(tmp_path):
    z, sd, _ = load_case("t0_b8")
    model = ArtifactModel(t0_params(), device=CPU, **P0_DIMS)
    model.load_state_dict(sd)
    path = tmp_path / "model.pt"
    model.save_model(path, artifact_log_priors=torch.tensor([1.0, 2.0]))
    saved = torch.load(path, weights_only=False)
    assert set(saved.keys()) == {"model_state_dict", "hyperparams", "num_read_features", "num_info_features",
                                 "ref_sequence_length", "artifact_log_priors", "artifact_spectra_state_dict"}
    assert type(saved["hyperparams"]).__module__ == "permutect.parameters"
    loaded, priors, spectra = load_model(path, device=CPU)
    assert spectra is None and torch.equal(priors, torch.tensor([1.0, 2.0]))
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1


def test_compute_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    z, sd, b = load_case("t0_b8")
    model = ArtifactModel(t0_params(), device=CPU, **P0_DIMS)
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"])
    with pytest.raises(L.PmtError):
        model.compute_batch_output(batch)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "permutect_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(pmt_[a-z_]+)\s*\(", header)))
    assert declared == sorted(L.EXPORTS)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pmt_abi_version() == L.ABI_VERSION


def test_ctypes_struct_sizes_match_the_compiled_library():
    lib = L.load()  # load() itself raises on any mismatch; spot-check two here
    assert lib.pmt_struct_bytes(0) == C.sizeof(L.PmtModel) and lib.pmt_struct_bytes(1) == C.sizeof(L.PmtBatch)
    assert lib.pmt_struct_bytes(99) < 0


def test_group_planner():
    rng = np.random.default_rng(0)
    ref = rng.integers(0, 11, size=5000).astype(np.int32)
    alt = rng.integers(1, 16, size=5000).astype(np.int32)
    from permutect_amd.data.batch import GroupPlan
    plan = GroupPlan(ref, alt)
    gs = plan.group_start
    assert gs[0] == 0 and gs[-1] == 5000 and np.all(np.diff(gs) > 0)
    tiles = 0
    for g in range(plan.num_groups):
        r, a = ref[gs[g]:gs[g + 1]].sum(), alt[gs[g]:gs[g + 1]].sum()
        tr, ta = (r + 15) // 16, (a + 15) // 16
        t = tr + ta
        # ref and alt tiles are dealt to disjoint waves, two tiles per wave
        assert (tr + 1) // 2 + (ta + 1) // 2 <= L.GROUP_WAVES and gs[g + 1] - gs[g] <= L.GROUP_MAX_SETS
        assert plan.group_tile_base[g] == tiles
        tiles += t
    assert plan.total_tiles == tiles
    # a set that cannot fit one group is refused loudly, naming the variant
    with pytest.raises(L.PmtError, match="variant 1"):
        GroupPlan(np.array([1, 200], dtype=np.int32), np.array([1, 200], dtype=np.int32))
    assert GroupPlan(np.zeros(0, np.int32), np.zeros(0, np.int32)).num_groups == 0


def test_batch_collate_matches_reference_layout():
    z, _, b = load_case("p0_b16")
    ints, floats, packed = b["int_array"], b["float_array"], b["packed_reads"]
    # per-datum construction (ref rows then alt rows per datum) collates to the reference's batch order
    nref, nalt = ints[:, 0].astype(int), ints[:, 1].astype(int)
    total_ref = nref.sum()
    ro, ao = np.concatenate([[0], np.cumsum(nref)]), np.concatenate([[0], np.cumsum(nalt)]) + total_ref
    data = [Datum(ints[i], floats[i], np.vstack([packed[ro[i]:ro[i + 1]], packed[ao[i]:ao[i + 1]]])) for i in range(len(ints))]
    batch = Batch(data)
    assert torch.equal(batch.packed_reads, torch.from_numpy(packed))
    assert batch.int_tensor.dtype == torch.int64 and batch.float_tensor.dtype == torch.float32
    np.testing.assert_array_equal(batch.get_reads_re().numpy(), z["reads_re_f16"])
    assert batch.get_reads_re().dtype == torch.float16 and batch.num_read_features() == 61
    oh = batch.get_one_hot_haplotypes_bcs()
    assert oh.shape == (16, 10, 21)
    hap = ints[:, 16:]
    assert oh[3, 2 * hap[3, 5] + 0, 5] == 1 and oh[3, 2 * hap[3, 21 + 5] + 1, 5] == 1  # refX / altX interleave


def test_decode_quirk_in_product_collate():
    z = np.load(f"{GOLDEN}/quirk_decode.npz")
    np.testing.assert_array_equal(decode_packed_reads(z["packed_reads"]), z["reads_re_f16"])


def test_downsampled_batch_reproduces_reference_gather_quirk():
    z = np.load(f"{GOLDEN}/quirk_downsample.npz")
    ref, alt = z["ref_counts"], z["alt_counts"]
    ints = np.zeros((3, 16 + 42), dtype=np.int16)
    ints[:, 0], ints[:, 1] = ref, alt
    floats = np.zeros((3, 6 + 71), dtype=np.float16)
    reads = np.random.default_rng(1).integers(0, 256, size=(int(ref.sum() + alt.sum()), 12), dtype=np.uint8)
    parent = Batch.from_arrays(ints, floats, reads)
    db = DownsampledBatch(parent, torch.ones(3), torch.ones(3))
    np.testing.assert_array_equal(db.read_indices.numpy(), z["read_indices_all_kept"])
    np.testing.assert_array_equal(db.get(Data.REF_COUNT).numpy(), z["new_ref_counts_all_kept"])
    np.testing.assert_array_equal(db.get(Data.ALT_COUNT).numpy(), z["new_alt_counts_all_kept"])
    fixed = DownsampledBatch(parent, torch.ones(3), torch.ones(3), fix_alt_gather=True)
    np.testing.assert_array_equal(fixed.read_indices.numpy(), np.arange(int(ref.sum() + alt.sum())))
    assert db.plan() is parent.plan()


def test_downsampler_bin_index_matches_reference_layout():
    """flattened (source, label, variant type, ref bin, alt bin) index and the count bins of reference
    data/count_binning.py:61-66 / data/batch.py:228-230."""
    import torch
    from permutect_amd.data.batch import Batch
    from permutect_amd.training import downsampler as D
    assert (D.NUM_REF_COUNT_BINS, D.NUM_ALT_COUNT_BINS) == (4, 5)
    assert D.ref_count_bin_indices(torch.tensor([0, 2, 3, 10, 40])).tolist() == [0, 0, 1, 3, 3]
    assert D.alt_count_bin_indices(torch.tensor([1, 3, 4, 15, 99])).tolist() == [0, 0, 1, 4, 4]
    ints = np.zeros((3, 58), dtype=np.int16)
    ints[:, 0], ints[:, 1] = [0, 7, 10], [1, 6, 15]          # ref / alt counts
    ints[:, 2], ints[:, 3], ints[:, 4] = [0, 1, 2], [0, 3, 4], [0, 0, 1]  # label, variant type, source
    floats = np.zeros((3, 77), dtype=np.float16)
    reads = np.zeros((int(ints[:, 0].sum() + ints[:, 1].sum()), 12), dtype=np.uint8)
    b = Batch.from_arrays(ints, floats, reads)
    idx = D.flattened_slvra_index(b).tolist()
    want = [(((s * 3 + l) * 5 + v) * 4 + r) * 5 + a for s, l, v, r, a in [(0, 0, 0, 0, 0), (0, 1, 3, 2, 1), (1, 2, 4, 3, 4)]]
    assert idx == want
    ds = D.Downsampler(num_sources=2)
    rf, af = ds.calculate_downsampling_fractions(b)
    assert rf.shape == (3,) and bool(((rf >= 0) & (rf <= 1)).all()) and bool(((af >= 0) & (af <= 1)).all())


def test_pack_order_fills_groups_and_keeps_every_variant():
    """pmt_pack_order: a permutation of the batch's variants under which the planner needs fewer groups (the WGS-shaped
    draw of the bench: 3643 -> 3414), with oversized variants (left to the split planner) kept."""
    import ctypes as C
    from permutect_amd.data.batch import GroupPlan, pack_order
    rng = np.random.default_rng(0)
    n = 65536
    ref, alt = rng.integers(0, 11, n), rng.integers(1, 16, n)
    order = pack_order(ref, alt)
    assert np.array_equal(np.sort(order), np.arange(n))
    before, after = GroupPlan(ref, alt).num_groups, GroupPlan(ref[order], alt[order]).num_groups
    assert before == 3643 and after <= 3420, (before, after)
    assert after * 256 >= int(ref.sum() + alt.sum())  # (nothing can beat full workgroups)
    ref[7], alt[4000] = 700, 900
    order = pack_order(ref, alt)
    assert np.array_equal(np.sort(order), np.arange(n))
    assert GroupPlan(ref[order], alt[order], allow_split=True).layered
    for tiny in (0, 1, 2):
        o = pack_order(ref[:tiny], alt[:tiny])
        assert np.array_equal(np.sort(o), np.arange(tiny))


def test_packed_batch_is_the_same_batch_reordered():
    z, sd, b = load_case("p0_deep")
    plain = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"])
    packed = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"], pack=True)
    o = packed.order
    assert o is not None and np.array_equal(np.sort(o), np.arange(plain.size()))
    assert torch.equal(packed.int_tensor, plain.int_tensor[o]) and torch.equal(packed.float_tensor, plain.float_tensor[o])
    # read rows: every variant keeps its own reads, ref block then alt block
    nref, nalt = plain.host_counts()
    r0 = np.concatenate([[0], np.cumsum(nref)]); a0 = int(nref.sum()) + np.concatenate([[0], np.cumsum(nalt)])
    pr, pa = packed.host_counts()
    pr0 = np.concatenate([[0], np.cumsum(pr)]); pa0 = int(pr.sum()) + np.concatenate([[0], np.cumsum(pa)])
    for i, v in enumerate(o):
        assert torch.equal(packed.packed_reads[pr0[i]:pr0[i + 1]], plain.packed_reads[r0[v]:r0[v + 1]])
        assert torch.equal(packed.packed_reads[pa0[i]:pa0[i + 1]], plain.packed_reads[a0[v]:a0[v + 1]])


def test_flat_parameter_space_is_laid_out_early_then_late():
    """engine/plan.py: ParamSpace orders the flat buffer [early | late] for the overlapped data-parallel reduction: every
    leaf whose gradient the per-variant kernels or the parametrization adjoint finish AFTER the read-set backward (info
    MLP, haplotype CNN, `.original` leaves) lies behind `late_start`; everything else in front of it."""
    from permutect_amd.engine.plan import ParamSpace
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=torch.device("cpu"), **P0_DIMS)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    space = ParamSpace(model, torch.device("cpu"))
    assert 0 < space.late_start < space.size
    for n, p in model.named_parameters():
        off = space.offset_of(p)
        late = n.startswith(("info_embedding.", "haplotypes_cnn.")) or n.endswith(".original")
        assert (off >= space.late_start) == late, n
        assert torch.equal(p.detach(), before[n])            # re-binding into the flat buffer keeps the values
        assert p.data_ptr() == space.theta.data_ptr() + 4 * off
    early_params = sum(p.numel() for n, p in model.named_parameters() if not ParamSpace.is_late(n))
    assert early_params > 0.75 * sum(p.numel() for p in model.parameters())  # the bucket that overlaps is the big one


def test_mean_loss_is_the_unweighted_mean_over_labels():
    """reference model_training.py:170 feeds the scheduler torch.mean(get_marginal(LABEL)): each label's average loss,
    averaged over labels -- not the pooled total / count."""
    from permutect_amd.training.loss_recorder import PRIMARY, LossRecorder
    rec = LossRecorder(torch.device("cpu"), num_sources=1)
    tot, cnt = rec.totals(PRIMARY), rec.counts(PRIMARY)
    tot[0, 0, 0, 1, 1], cnt[0, 0, 0, 1, 1] = 30.0, 10.0      # label 0: average 3
    tot[0, 0, 1, 2, 0], cnt[0, 0, 1, 2, 0] = 10.0, 10.0      # label 0 again: pooled average (30 + 10) / 20 = 2
    tot[0, 1, 0, 0, 0], cnt[0, 1, 0, 0, 0] = 8.0, 1.0        # label 1: average 8 on a single variant
    assert abs(rec.mean_loss() - (2.0 + 8.0) / 2) < 1e-6       # label 2 has no data: left out (the reference gives NaN)
    tot[0, 2, 0, 0, 0], cnt[0, 2, 0, 0, 0] = 5.0, 5.0
    assert abs(rec.mean_loss() - (2.0 + 8.0 + 1.0) / 3) < 1e-6


def test_split_planner_streams_rows_into_full_groups():
    """pmt_plan_groups_split with oversized sets: every ref / alt row lies in exactly one group, a group's rows are one
    contiguous run per side that fits the wave layout (ref tiles and alt tiles on disjoint waves, two per wave), its variant
    range covers exactly the sets it holds rows of, and the groups are full: ~tiles / 16 of them, not a partly empty last
    group per oversized set."""
    from permutect_amd.data.batch import GroupPlan
    from permutect_amd.engine import lib as L
    rng = np.random.default_rng(3)
    for ref, alt in [(rng.poisson(300, 200), np.maximum(rng.poisson(300, 200), 1)),          # the stress shape
                     (rng.integers(0, 11, 500), rng.integers(1, 16, 500)),                    # WGS-shaped ...
                     (np.array([0, 700, 3, 0, 256, 5]), np.array([1, 1, 2, 600, 1, 9]))]:     # ... and a ragged mix
        ref, alt = ref.astype(np.int64), alt.astype(np.int64)
        if ref.max() + alt.max() < 200:
            ref[17], alt[400] = 290, 310  # two oversized sets among small ones
        plan = GroupPlan(ref, alt, allow_split=True)
        assert plan.layered
        span = plan.span[: plan.num_groups]
        ro, ao = np.concatenate([[0], np.cumsum(ref)]), np.concatenate([[0], np.cumsum(alt)])
        rpos = apos = 0
        for v0, v1, rb, re, ab, ae in span:
            assert rb == rpos and ab == apos and re >= rb and ae >= ab and (re - rb) + (ae - ab) > 0
            rpos, apos = re, ae
            tr, ta = -(-(re - rb) // 16), -(-(ae - ab) // 16)
            assert -(-tr // 2) + -(-ta // 2) <= L.GROUP_WAVES
            assert 0 < v1 - v0 <= L.GROUP_MAX_SETS
            held = [v for v in range(len(ref)) if (min(re, ro[v + 1]) > max(rb, ro[v])) or (min(ae, ao[v + 1]) > max(ab, ao[v]))]
            assert held and held[0] == v0 and held[-1] == v1 - 1
        assert rpos == ref.sum() and apos == alt.sum()
        tiles = int(np.diff(plan.group_tile_base[: plan.num_groups + 1]).sum())
        assert plan.total_tiles == tiles
        if len(ref) == 200:  # 600-read sets: 38 tiles each used to take 3 groups (48 tile slots)
            assert plan.num_groups <= 1.08 * tiles / L.GROUP_TILES + 1, (plan.num_groups, tiles)


def test_kernel_instances_are_chosen_by_the_models_tile_shape(monkeypatch):
    """engine/instances.py: which build of the library runs a model.  Host logic only -- the descriptor is lowered
    on the CPU and `pmt_shape_id` is a host function of every build: P0 fills the default library's shape with its widths compiled in
    (2); the reference's test configuration T0 fills no tile of it (0 there) but has an exact shape of its own, for which the table's
    prebuilt library answers 6 (its tiles, widths at run time); a model with the production TILES but other widths runs the default
    library's tile-exact 16-bit instances (6) instead of the generic one; a read MLP that starts with a skip block has no exact shape."""
    import ctypes as C
    from permutect_amd.engine import instances as I
    from permutect_amd.engine import lib as L
    from permutect_amd.engine.plan import EnginePlan, ParamSpace
    from permutect_amd.parameters import t0_params
    monkeypatch.delenv("PMT_SHAPE", raising=False)
    monkeypatch.delenv("PMT_LIB", raising=False)
    monkeypatch.setenv("PMT_JIT", "0")  # (this test must not start a two-minute build)

    def desc_of(params):
        model = ArtifactModel(params, device=torch.device("cpu"), **P0_DIMS)
        return EnginePlan(model, ParamSpace(model, torch.device("cpu")), torch.device("cpu")).desc
    default = L.load()
    assert L.shape_of(default) == (4, 2, 4, 1, 61, 30, 60, 10, 10)
    p0 = desc_of(p0_params())
    assert default.pmt_shape_id(C.byref(p0)) == 2 and I.exact_shape_of(p0) == (4, 2, 4, 1, 61, 30, 60, 10, 10)
    assert I.library_for(p0) is default
    t0 = desc_of(t0_params())
    assert default.pmt_shape_id(C.byref(t0)) == 0 and I.exact_shape_of(t0) == (4, 1, 2, 2, 61, 10, 30, 10, 20)
    t0_lib = os.path.join(I.INSTANCE_DIR, "libpermutect_amd_4_1_2_2_61_10_30_10_20.so")
    if os.path.exists(t0_lib):  # built by __graft_entry__.build() (make instances)
        lib = I.library_for(t0)
        assert lib is not default and L.shape_of(lib)[:4] == (4, 1, 2, 2) and lib.pmt_shape_id(C.byref(t0)) == 6
    other = p0_params()
    other.read_layers, other.info_layers = [24, -2], [24, -1]  # D = 24 + 24 + 10 = 58: the production tiles, other widths
    od = desc_of(other)
    assert I.exact_shape_of(od)[:4] == (4, 2, 4, 1) and default.pmt_shape_id(C.byref(od)) == 6 and I.library_for(od) is default
    odd = p0_params()
    odd.read_layers = [-2, 30]  # (the read MLP starts with a skip block over the 61 read features: no tile-exact instance)
    with pytest.warns(UserWarning, match="GENERIC"):
        assert I.exact_shape_of(desc_of(odd)) is None and I.library_for(desc_of(odd)) is default
    # layers wider than 64: the wide build of the library (pmt_limits), the default one refuses them; beyond 128 nothing runs
    from permutect_amd.parameters import wide_params
    assert L.limits_of(default) == {"max_width": 64, "max_half_ffn": 16, "slot_floats": 1024, "group_waves": 8}
    wide_exact = os.path.join(I.INSTANCE_DIR, "libpermutect_amd_4_3_7_2_61_48_98_16_20.so")
    if os.path.exists(I.WIDE_LIB) and os.path.exists(wide_exact):  # built by __graft_entry__.build() (make wide, make instances)
        wd = desc_of(wide_params())
        assert I.widest_layer(wd) == 98 and wd.d_model == 98 and I.exact_shape_of(wd) == (4, 3, 7, 2, 61, 48, 98, 16, 20)
        lib = I.library_for(wd)  # the exact instances built around this shape, with 8-tile register arrays
        assert L.limits_of(lib)["max_width"] == 128 and L.shape_of(lib) == (4, 3, 7, 2, 61, 48, 98, 16, 20) and lib.pmt_shape_id(C.byref(wd)) == 2
        assert lib.pmt_model_check(C.byref(wd)) == 0 and default.pmt_model_check(C.byref(wd)) == L.E_UNSUPPORTED
        wd.force_shape = 2  # (PMT_SHAPE=any; also what a wide model without an exact shape gets): the generic instances of the wide build
        with pytest.warns(UserWarning, match="WIDE build"):
            wide = I.library_for(wd)
        assert L.limits_of(wide) == {"max_width": 128, "max_half_ffn": 16, "slot_floats": 2048, "group_waves": 8}
        assert wide is not lib and wide.pmt_shape_id(C.byref(wd)) == 0 and wide.pmt_model_check(C.byref(wd)) == 0
    # d_ffn / 2 beyond 16: builds with two 16-feature tiles per half of the gated blocks' hidden layer (pmt_limits: max_half_ffn 32) -- the
    # exact instances around the production widths with d_ffn 48 (make instances), the wide32 build for everything else
    half = p0_params()
    half.self_attention_hidden_dimension = 48
    p0h_lib = os.path.join(I.INSTANCE_DIR, "libpermutect_amd_4_2_4_1_61_30_60_24_10.so")
    if os.path.exists(p0h_lib) and os.path.exists(I.WIDE32_LIB):
        hd = desc_of(half)
        assert I.half_tiles(hd) == 2 and I.exact_shape_of(hd) == (4, 2, 4, 1, 61, 30, 60, 24, 10) and default.pmt_model_check(C.byref(hd)) == L.E_UNSUPPORTED
        lib = I.library_for(hd)
        assert L.limits_of(lib) == {"max_width": 64, "max_half_ffn": 32, "slot_floats": 1024, "group_waves": 8} and lib.pmt_shape_id(C.byref(hd)) == 2
        from permutect_amd.parameters import wide64_params
        with pytest.warns(UserWarning, match="WIDE build"):  # (no exact instances prebuilt for this shape, and PMT_JIT = 0 here)
            w64 = desc_of(wide64_params())
            lib64 = I.library_for(w64)
        assert L.limits_of(lib64)["max_half_ffn"] == 32 and L.limits_of(lib64)["max_width"] == 128 and lib64.pmt_shape_id(C.byref(w64)) == 0
        assert I.wide_library().pmt_model_check(C.byref(w64)) == L.E_UNSUPPORTED  # (the one-tile wide build does not take it)
    too_wide = wide_params()
    too_wide.read_layers = [130]
    with pytest.raises(L.PmtError, match="exceeds"):
        desc_of(too_wide)


def test_no_mfma_accumulates_onto_another_opcodes_result_without_wait_states():
    """Round 4's "race" was a hardware hazard the compiler does not know: a `v_mfma_f32_16x16x16_bf16 D, a, b, D` right behind a
    `v_mfma_f32_16x16x32_bf16` that writes D reads a stale accumulator (profiles/r05_mfma_chain_hazard.txt).  The kernels no longer chain
    the two shapes; this test keeps it that way for every library the build made: scripts/check_mfma_chains.py disassembles their gfx950
    code and fails on any MFMA whose SrcC is the result of an MFMA of another opcode issued fewer than ten issue slots before."""
    import glob
    import importlib.util
    from concurrent.futures import ThreadPoolExecutor
    spec = importlib.util.spec_from_file_location("check_mfma_chains", os.path.join(ROOT, "scripts", "check_mfma_chains.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    # the scanner itself, on the two shapes of the hazard and on what must pass
    bad = """0000000000001000 <kernel_a>:
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]
\ts_nop 1
\tv_mfma_f32_16x16x16_bf16 v[0:3], v[16:17], v[18:19], v[0:3]
"""
    good = """0000000000001000 <kernel_b>:
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]
\tv_mfma_f32_16x16x16_bf16 v[4:7], v[16:17], v[18:19], v[4:7]
\ts_nop 7
\ts_nop 1
\tv_mfma_f32_16x16x16_bf16 v[0:3], v[16:17], v[18:19], v[0:3]
"""
    assert len(chk.scan(bad)) == 1 and chk.scan(bad)[0][1:] == ("v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x16_bf16", 2)
    assert chk.scan(good) == []
    libs = sorted(glob.glob(os.path.join(ROOT, "permutect_amd", "*.so")) + glob.glob(os.path.join(ROOT, "permutect_amd", "instances", "*.so")))
    assert len(libs) >= 5, libs  # the default library, alt6, the wide builds, the per-shape instances
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        found = [f for per_lib in pool.map(lambda p: chk.check([p]), libs) for f in per_lib]
    assert found == [], found[:5]


def test_instance_libraries_carry_the_default_librarys_build_id(tmp_path, monkeypatch):
    """a per-shape library kept from an earlier state of the tree must not be reused.  Every build of the library carries a hash
    of the sources (pmt_build_id); engine/instances.py reads it out of the FILE (a library it has mapped cannot be replaced by a rebuilt
    one) and takes only libraries whose id is the default library's."""
    import glob
    import shutil
    import warnings
    from permutect_amd.engine import instances as I
    default_id = L.build_id(L.load())
    assert re.fullmatch(r"[0-9a-f]{16}", default_id) and default_id != "0" * 16
    libs = sorted(glob.glob(os.path.join(I.INSTANCE_DIR, "*.so")))
    assert libs and all(I.file_build_id(p) == default_id and I.is_current(p) for p in libs)
    # a library from other sources: the same file with another id
    stale = tmp_path / os.path.basename(libs[0])
    blob = open(libs[0], "rb").read()
    assert blob.count(b"PMT_BUILD_ID=" + default_id.encode()) >= 1
    stale.write_bytes(blob.replace(b"PMT_BUILD_ID=" + default_id.encode(), b"PMT_BUILD_ID=" + b"0123456789abcdef"))
    assert I.file_build_id(str(stale)) == "0123456789abcdef" and not I.is_current(str(stale))
    assert I.file_build_id(str(tmp_path / "missing.so")) is None
    # ... is ignored by library_for (PMT_JIT=0: nothing is rebuilt; the model runs the generic instance, loudly)
    monkeypatch.setenv("PMT_JIT", "0")
    monkeypatch.setenv("PMT_INSTANCE_DIR", str(tmp_path))
    monkeypatch.setattr(I, "INSTANCE_DIR", str(tmp_path / "nothing_here"))
    model = ArtifactModel(t0_params(), device=CPU, **P0_DIMS)
    from permutect_amd.engine.plan import EnginePlan, ParamSpace
    desc = EnginePlan(model, ParamSpace(model, CPU), CPU).desc
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        lib = I.library_for(desc)
    assert lib is L.load() and any("OTHER sources" in str(w.message) for w in caught), [str(w.message)[:80] for w in caught]
