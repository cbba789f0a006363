"""Device-side read downsampling (pmt_downsample_counts / _index) against the reference's scheme (data/batch.py:389-439,
training/downsampler.py:105-123): structural invariants exactly, the random choices in distribution."""
import numpy as np
import pytest
import torch

from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch, DownsampledBatch
from permutect_amd.parameters import P0_DIMS, p0_params
from tests.helpers import load_case

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _batch(nvar=2000, seed=0):
    rng = np.random.default_rng(seed)
    nref, nalt = rng.integers(0, 11, nvar), rng.integers(1, 16, nvar)
    ints = np.zeros((nvar, 58), dtype=np.int16)
    ints[:, 0], ints[:, 1] = nref, nalt
    ints[:, 2] = np.arange(nvar) % 3
    ints[:, 16:] = rng.integers(0, 5, (nvar, 42))
    floats = np.zeros((nvar, 77), dtype=np.float16)
    floats[:, 6:] = rng.standard_normal((nvar, 71)).astype(np.float16)
    packed = rng.integers(0, 256, (int(nref.sum() + nalt.sum()), 12), dtype=np.uint8)
    return Batch.from_arrays(ints, floats, packed).copy_to(DEV), nref, nalt


def _check_structure(db, nref, nalt, fix):
    total_ref = int(nref.sum())
    rc, ac = db.ref_counts.cpu().numpy(), db.alt_counts.cpu().numpy()
    assert np.all(rc <= nref) and np.all(ac <= nalt) and np.all(ac >= 1)
    idx = db.read_indices.cpu().numpy()
    kept_ref, kept_alt = idx[: rc.sum()], idx[rc.sum(): rc.sum() + ac.sum()]
    ref_start = np.concatenate([[0], np.cumsum(nref)])
    alt_start = np.concatenate([[0], np.cumsum(nalt)])
    # ascending kept rows, each inside its own variant's range, in variant order
    assert np.all(np.diff(kept_ref) > 0) and np.all(np.diff(kept_alt) > 0)
    owner = np.repeat(np.arange(len(nref)), rc)
    assert np.all(kept_ref >= ref_start[owner]) and np.all(kept_ref < ref_start[owner + 1])
    owner = np.repeat(np.arange(len(nalt)), ac)
    alt_local = kept_alt - (total_ref if fix else 0)  # the reference gathers alt rows un-offset (quirk), fix adds total_ref
    assert np.all(alt_local >= alt_start[owner]) and np.all(alt_local < alt_start[owner + 1])


@pytest.mark.parametrize("fix", [False, True])
def test_given_fractions_structure_and_rates(fix):
    batch, nref, nalt = _batch()
    rf = torch.full((batch.size(),), 0.3, device=DEV)
    af = torch.full((batch.size(),), 0.7, device=DEV)
    db = DownsampledBatch.on_device(batch, seed=11, ref_fracs_b=rf, alt_fracs_b=af, fix_alt_gather=fix, force_random=37)
    _check_structure(db, nref, nalt, fix)
    rc, ac = db.ref_counts.cpu().numpy(), db.alt_counts.cpu().numpy()
    assert abs(rc.sum() / nref.sum() - 0.3) < 0.02          # Bernoulli(0.3) over ~10 000 reads
    expect_alt = (0.7 * (nalt - 1) + 1).sum()                # one forced read + Bernoulli(0.7) on the others
    assert abs(ac.sum() / expect_alt - 1.0) < 0.02
    # same seed -> same decisions; another seed -> different
    db2 = DownsampledBatch.on_device(batch, seed=11, ref_fracs_b=rf, alt_fracs_b=af, fix_alt_gather=fix, force_random=37)
    assert torch.equal(db.ref_counts, db2.ref_counts) and torch.equal(db.read_indices[: int(rc.sum())], db2.read_indices[: int(rc.sum())])
    db3 = DownsampledBatch.on_device(batch, seed=12, ref_fracs_b=rf, alt_fracs_b=af, fix_alt_gather=fix, force_random=37)
    assert not torch.equal(db.ref_counts, db3.ref_counts)


def test_fractions_follow_the_beta_mixture():
    batch, nref, nalt = _batch(nvar=20000, seed=1)
    db = DownsampledBatch.on_device(batch, seed=5)
    f = db.ref_fracs.cpu().numpy()
    assert f.min() >= 0.0 and f.max() <= 1.0
    # uniform mixture of Beta(1,1), Beta(1,5), Beta(5,1), Beta(5,5): mean 1/2, variance = mean of (var_k + mean_k^2) - 1/4
    means = np.array([0.5, 1 / 6, 5 / 6, 0.5])
    var = np.array([1 / 12, 5 / (36 * 7), 5 / (36 * 7), 25 / (100 * 11)])
    assert abs(f.mean() - means.mean()) < 0.01
    assert abs(f.var() - ((var + means ** 2).mean() - means.mean() ** 2)) < 0.01
    # component weights steer the mixture: all weight on Beta(5,1) -> mean 5/6
    w = torch.zeros(batch.size(), 4, device=DEV)
    w[:, 2] = 1.0
    db = DownsampledBatch.on_device(batch, seed=6, ref_weights_b4=w, alt_weights_b4=w)
    assert abs(float(db.ref_fracs.mean()) - 5 / 6) < 0.01 and abs(float(db.alt_fracs.mean()) - 5 / 6) < 0.01


def test_all_kept_equals_parent_and_matches_torch_path():
    z, sd, b = load_case("p0_b16")
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=DEV, **P0_DIMS)
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(DEV)
    ones = torch.ones(batch.size(), device=DEV)
    db = DownsampledBatch.on_device(batch, seed=3, ref_fracs_b=ones, alt_fracs_b=ones, fix_alt_gather=True)
    ref = DownsampledBatch(batch, ones, ones, fix_alt_gather=True)
    n = ref.read_indices.numel()
    assert torch.equal(db.read_indices[:n], ref.read_indices) and torch.equal(db.ref_counts, ref.ref_counts)
    with torch.no_grad():
        a = model.compute_batch_output(batch)
        c = model.compute_batch_output(db)
    assert torch.allclose(a.logits_b, c.logits_b, rtol=1e-5, atol=1e-5)
    # and the reference-quirk mode reproduces the reference's index for "keep everything"
    quirk = DownsampledBatch.on_device(batch, seed=3, ref_fracs_b=ones, alt_fracs_b=ones)
    ref_q = DownsampledBatch(batch, ones, ones)
    assert torch.equal(quirk.read_indices[:n], ref_q.read_indices)


def test_downsampler_drives_the_fused_kernels_in_a_train_step():
    """Downsampler.downsample (bin lookup of the mixture weights + two launches) feeding one full training step."""
    from permutect_amd.training.downsampler import Downsampler
    from permutect_amd.training.optimizer import FusedClipAdamW, backpropagate
    batch, nref, nalt = _batch(nvar=512, seed=4)
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=DEV, **P0_DIMS)
    model.train(True)
    ds = Downsampler(num_sources=1).to(DEV)
    with torch.no_grad():  # all weight on Beta(1,5) for SNV artifacts: their reads are thinned hardest
        ds.parametrizations.log_ref_weights_slvrak.original[0, 0, 0, :, :, 1] = 20.0
    db = ds.downsample(batch, seed=9, fix_alt_gather=True)
    _check_structure(db, nref, nalt, True)
    labels = batch.get(Data_LABEL()).cpu().numpy()
    fr = db.ref_fracs.cpu().numpy()
    assert fr[labels == 0].mean() < 0.25 < fr[labels != 0].mean()
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    out = model.compute_batch_output(db)
    losses = model.compute_batch_losses(out, db)
    backpropagate(opt, losses.total_loss, params_to_clip=model.parameters())
    torch.cuda.synchronize()
    assert torch.isfinite(losses.total_loss) and float(opt.grad_norm.item()) > 0


def test_downsampling_a_batch_with_read_sets_beyond_one_workgroup():
    """A split plan names explicit rows, so a DownsampledBatch of such a parent plans from its own counts: keeping every
    read reproduces the parent's outputs, and a thinned batch runs a finite training step."""
    from permutect_amd.training.optimizer import FusedClipAdamW, backpropagate
    from tests.test_forward_gpu import _arrays
    nref = np.array([5, 300, 0, 10, 700, 2, 256, 9])
    nalt = np.array([3, 350, 600, 15, 1, 7, 255, 1])
    ints, floats, packed = _arrays(nref, nalt, seed=21)
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=DEV, **P0_DIMS)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(DEV)
    ones = torch.ones(batch.size(), device=DEV)
    db = DownsampledBatch.on_device(batch, seed=3, ref_fracs_b=ones, alt_fracs_b=ones, fix_alt_gather=True)
    with torch.no_grad():
        a = model.compute_batch_output(batch)
        c = model.compute_batch_output(db)
    assert batch.plan(allow_split=True).layered and db.plan(allow_split=True).layered
    assert torch.allclose(a.logits_b, c.logits_b, rtol=1e-5, atol=1e-5)
    model.train(True)
    half = 0.5 * ones
    thin = DownsampledBatch.on_device(batch, seed=4, ref_fracs_b=half, alt_fracs_b=half, fix_alt_gather=True)
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    out = model.compute_batch_output(thin)
    losses = model.compute_batch_losses(out, thin)
    backpropagate(opt, losses.total_loss, params_to_clip=model.parameters())
    torch.cuda.synchronize()
    assert torch.isfinite(losses.total_loss) and float(opt.grad_norm.item()) > 0


def test_device_kernels_reproduce_the_reference_quirk_fixture():
    """tests/golden/quirk_downsample.npz was written by the REFERENCE's DownsampledBatch (data/batch.py:389-439): with every
    read kept its `read_indices` are all ref rows, then the alt rows WITHOUT the offset of the ref region.
    pmt_downsample_counts / _index in reference mode must give exactly that index and those counts; in fixed mode the alt
    rows carry the offset."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "quirk_downsample.npz"))
    nref, nalt = z["ref_counts"], z["alt_counts"]
    ints = np.zeros((len(nref), 58), dtype=np.int16)
    ints[:, 0], ints[:, 1] = nref, nalt
    floats = np.zeros((len(nref), 77), dtype=np.float16)
    packed = np.zeros((int(nref.sum() + nalt.sum()), 12), dtype=np.uint8)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(DEV)
    ones = torch.ones(batch.size(), device=DEV)
    want = z["read_indices_all_kept"]
    quirk = DownsampledBatch.on_device(batch, seed=1, ref_fracs_b=ones, alt_fracs_b=ones)
    assert np.array_equal(quirk.read_indices[: len(want)].cpu().numpy(), want)
    assert np.array_equal(quirk.ref_counts.cpu().numpy(), z["new_ref_counts_all_kept"])
    assert np.array_equal(quirk.alt_counts.cpu().numpy(), z["new_alt_counts_all_kept"])
    fixed = DownsampledBatch.on_device(batch, seed=1, ref_fracs_b=ones, alt_fracs_b=ones, fix_alt_gather=True)
    tr = int(nref.sum())
    assert np.array_equal(fixed.read_indices[: len(want)].cpu().numpy(), np.concatenate([want[:tr], want[tr:] + tr]))
    # the reference's half-kept draw has the same STRUCTURE (its random stream is torch's, not reproducible on the device):
    # kept rows ascending per side, counts within bounds, at least one alt read per variant
    rc, ac, idx = z["new_ref_counts_half"], z["new_alt_counts_half"], z["read_indices_half"]
    assert np.all(rc <= nref) and np.all(ac <= nalt) and np.all(ac >= 1) and len(idx) == rc.sum() + ac.sum()
    half = DownsampledBatch.on_device(batch, seed=2, ref_fracs_b=0.5 * ones, alt_fracs_b=0.5 * ones)
    _check_structure(half, nref, nalt, False)


@pytest.mark.parametrize("component", [0, 1, 2, 3])
def test_keep_fractions_follow_each_beta_component(component):
    """Kolmogorov-Smirnov test of the device's keep fractions against the Beta shape of every mixture component
    (reference training/downsampler.py:27,118-122: Beta(1,1), Beta(1,5), Beta(5,1), Beta(5,5)), ref and alt."""
    from scipy import stats
    batch, _, _ = _batch(nvar=20000, seed=3 + component)
    w = torch.zeros(batch.size(), 4, device=DEV)
    w[:, component] = 1.0
    db = DownsampledBatch.on_device(batch, seed=40 + component, ref_weights_b4=w, alt_weights_b4=w)
    a, b = [(1.0, 1.0), (1.0, 5.0), (5.0, 1.0), (5.0, 5.0)][component]
    for f in (db.ref_fracs.cpu().numpy(), db.alt_fracs.cpu().numpy()):
        d, pval = stats.kstest(f.astype(np.float64), stats.beta(a, b).cdf)
        assert d < 0.02 and pval > 1e-4, (component, d, pval)  # n = 20 000: the 1e-4 critical distance is ~0.016
    # the uniform mixture: KS against the mixture's cdf
    db = DownsampledBatch.on_device(batch, seed=77)
    mix = lambda x: 0.25 * sum(stats.beta(p, q).cdf(x) for p, q in [(1, 1), (1, 5), (5, 1), (5, 5)])  # noqa: E731
    d, pval = stats.kstest(db.ref_fracs.cpu().numpy().astype(np.float64), mix)
    assert d < 0.02 and pval > 1e-4, (d, pval)


def Data_LABEL():
    from permutect_amd.data.datum import Data
    return Data.LABEL
