"""The one-launch fit of short tensors in the reference's own arithmetic (run with -m gpu): nnc_kmeans_fit_reference_f32
(include/nnc.h) = KMeans(n_clusters=k, init=space, n_init=1, algorithm="full").fit (utility.py:237-238) with scikit-learn's
float32 running sums in sample order.  Checked against the oracle's mode A (the restatement pinned on the reference's own
outputs, tests/test_oracle.py) with the device's documented relocation order, bit for bit: n_iter_, every centre, every
index -- and against the reference's outputs themselves in tests/test_gpu_parity.py / test_gpu_kmeanspp.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from neural_network_compression_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def km():
    assert torch.cuda.is_available()
    from neural_network_compression_amd import _native, kmeans

    _native.load()
    return kmeans


def _check(km, x, init, want_values=True):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    model, vals = km.fit_reference(t, init, want_values=want_values)
    oa = orc.kmeans_lloyd(x, init, accum="A", reloc="descending")
    assert model.n_iter_ == oa.n_iter_, (model.n_iter_, oa.n_iter_)
    assert np.array_equal(model.cluster_centers_.ravel(), oa.cluster_centers_.ravel())
    assert np.array_equal(model.labels_, oa.labels_), int((model.labels_ != oa.labels_).sum())
    assert model.stop_reason_ == ("strict" if oa.strict else ("tol" if oa.n_iter_ < 300 else model.stop_reason_))
    if want_values:
        assert np.array_equal(vals.cpu().numpy(), oa.cluster_centers_.ravel()[oa.labels_])
    else:
        assert vals is None
    assert np.array_equal(model.counts_device_.cpu().numpy(), np.bincount(oa.labels_, minlength=len(init)))
    assert model.n_relocations_ == oa.reloc_info_.get("reloc_events", 0)
    assert model.n_reloc_multi_ == oa.reloc_info_.get("reloc_multi", 0)
    return model, oa


@pytest.mark.parametrize("n,k", [(4, 4), (5, 4), (17, 16), (33, 32), (63, 7), (64, 8), (65, 9), (127, 16), (128, 16), (129, 33),
                                 (300, 4), (768, 16), (1000, 65), (1023, 32), (1024, 128), (2304, 16), (3072, 16), (4095, 64),
                                 (4096, 16), (4096, 128)])
def test_sizes_linear_init(km, n, k):
    x = synth.weights((n,), 4200 + n + k)
    _check(km, x, np.linspace(x.min(), x.max(), k).astype(np.float32))


@pytest.mark.parametrize("seed", range(6))
def test_pruned_forgy_and_density(km, seed):
    rs = np.random.RandomState(seed)
    n = int(rs.randint(40, 4097))
    x = synth.weights((n,), 5200 + seed)
    orc.prune_weigth(x, 1.0, True)                       # many exact zeros, a gap around them
    k = int(rs.choice([4, 16, 32]))
    _check(km, x, x[rs.randint(0, n, size=k)], want_values=bool(seed & 1))   # forgy: duplicate initial centres -> relocations
    nzv = x[x != 0]
    cdfs = orc.get_weight_distribution(nzv)
    _check(km, x, orc.init_space(x, 4, "density", cdfs))


def test_relocation_paths(km):
    """Duplicate initial centres and centres in the pruned gap: empty clusters every way."""
    x = synth.weights((4000,), 9002)
    x[np.abs(x) < 0.06] = 0
    for init in [np.linspace(x.min(), x.max(), 16).astype(np.float32),
                 np.array([0.0, 0.0, 0.0, 0.1, 0.1, -0.1, 0.05, 0.0], dtype=np.float32),
                 np.zeros(40, dtype=np.float32)]:
        model, oa = _check(km, x, init)
        assert model.n_relocations_ >= 1
    # all samples equal: relocation bails out (max distance 0), empty centres copy the biggest (its raw sum if it comes later)
    xe = np.full(64, 0.125, dtype=np.float32)
    for init in ([0.125, 0.5, -0.5, 0.125], [0.5, -0.5, 0.125, 0.125]):
        model, _ = _check(km, xe, np.array(init, dtype=np.float32))
        assert model.n_relocations_ == 0


def test_tie_at_the_cut_is_reported(km):
    # two samples at the same distance from the only non-empty centre, one empty cluster: which one moves is numpy's choice
    x = np.array([-1.0, 1.0, 0.0, 0.0, 0.0, 0.0], dtype=np.float32)
    init = np.array([0.0, 0.0], dtype=np.float32)
    model, oa = _check(km, x, init)
    assert model.reloc_tie_ >= 1 and oa.reloc_info_.get("reloc_ties", 0) >= 1


def test_arguments(km):
    x = torch.from_numpy(synth.weights((5000,), 1)).cuda()
    with pytest.raises(ValueError):
        km.fit_reference(x, np.zeros(4, dtype=np.float32))            # longer than NNC_REF_NMAX
    with pytest.raises(ValueError):
        km.fit_reference(x[:100], np.zeros(129, dtype=np.float32))    # more centres than NNC_REF_KMAX (and than samples)
    # beyond the one-launch form the same arithmetic runs step by step (fit_reference_large): against the oracle's mode A
    init = np.linspace(-0.1, 0.1, 4).astype(np.float32)
    model, vals = km.fit_vector(x, init, arith="reference")
    assert model.arith_ == "reference"
    oa = orc.kmeans_lloyd(x.cpu().numpy(), init, accum="A")
    assert model.n_iter_ == oa.n_iter_
    assert np.array_equal(model.cluster_centers_.ravel().view(np.uint32), oa.cluster_centers_.ravel().view(np.uint32))
    assert np.array_equal(model.labels_, oa.labels_)
    assert np.array_equal(vals.cpu().numpy(), oa.cluster_centers_.ravel()[oa.labels_])
    model, _ = km.fit_vector(x, np.linspace(-0.1, 0.1, 4).astype(np.float32), arith="auto")
    assert model.arith_ == "fixed"
    model, _ = km.fit_vector(x[:4096], np.linspace(-0.1, 0.1, 4).astype(np.float32), arith="auto")
    assert model.arith_ == "reference"
    model, _ = km.fit_vector(x[:4096], np.linspace(-0.1, 0.1, 4).astype(np.float32), arith="fixed")
    assert model.arith_ == "fixed"


def test_auto_and_fixed_agree_to_summation_error(km):
    x = synth.weights((3000,), 77)
    t = torch.from_numpy(x).cuda()
    init = np.linspace(x.min(), x.max(), 16).astype(np.float32)
    a, _ = km.fit_vector(t, init, arith="reference")
    b, _ = km.fit_vector(t, init, arith="fixed")
    assert a.n_iter_ == b.n_iter_
    assert np.max(np.abs(a.cluster_centers_ - b.cluster_centers_)) <= 1e-6 * np.abs(b.cluster_centers_).max()
