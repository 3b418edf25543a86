"""Centroid fine-tuning (described but not implemented by the reference, papers/lat/report.tex:149-158):
the per-centroid gradient sum and the decode step on the GPU (run with -m gpu)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from neural_network_compression_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    from neural_network_compression_amd import _native, kmeans, ops
    from neural_network_compression_amd.common import trainer

    _native.load()
    return ops, kmeans, trainer


@pytest.mark.parametrize("n,k", [(1, 1), (1000, 4), (30_000, 16), (235_200, 33), (1_000_003, 256), (400_000, 257), (50_000, 1025)])
def test_centroid_gradient_is_the_exact_fixed_point_sum(mods, n, k):
    ops, _, _ = mods
    rng = np.random.RandomState(n % 97)
    g = (rng.randn(n) * 1e-3).astype(np.float32)
    g[::7] = 0
    labels = rng.randint(0, k, size=n)
    lab_t = torch.from_numpy(labels.astype(np.uint8 if k <= 256 else np.int16)).cuda()
    out = ops.centroid_gradient(torch.from_numpy(g).cuda(), lab_t, k).cpu().numpy()
    S = ops.fix_shift(float(np.abs(g).max()), n)
    q = np.rint(np.ldexp(g.astype(np.float64), S)).astype(np.int64)
    want = np.zeros(k, dtype=np.int64)
    np.add.at(want, labels, q)
    assert np.array_equal(out, np.ldexp(want.astype(np.float64), -S))
    # and it is the float64 sum of the gradients to the fixed point's resolution
    ref = np.bincount(labels, weights=g.astype(np.float64), minlength=k)
    cnt = np.bincount(labels, minlength=k)
    assert np.all(np.abs(out - ref) <= cnt * 2.0 ** (-S - 1) + 1e-30)
    # decode
    cen = rng.randn(k).astype(np.float32)
    dec = ops.gather(torch.from_numpy(cen).cuda(), lab_t).cpu().numpy()
    assert np.array_equal(dec, cen[labels])


def test_fine_tune_centroids_keeps_the_indices_and_lowers_the_loss(mods):
    ops, _, tr = mods
    from neural_network_compression_amd import le_net_300_100_trainer as lt

    tr.Trainer.pruned_indexes_by_layer.clear()
    torch.manual_seed(0)
    t = lt.LeNet300100Trainer()
    for li, (name, wshape, bshape) in enumerate(synth.LENET_300_100):
        layer = getattr(t.neural_network, name)
        layer.set_weights([torch.from_numpy(synth.weights(wshape, 2000 + 2 * li)).cuda(), torch.from_numpy(synth.weights(bshape, 2001 + 2 * li)).cuda()])
    rng = np.random.RandomState(1)
    x = rng.rand(2048, 784).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.randint(0, 10, size=2048)]
    data = tr.LeNetDataset(x, y)
    test = tr.LeNetDataset(x[:256], y[:256].argmax(1))
    t._prune_parameters(True)
    t.quantize(test, False, 4, "linear")
    before = {}
    for layer, ms in t.quantized_models_by_layer.items():
        before[layer] = [w.clone() for w in layer.get_weights()]
        for w, m in zip(layer.get_weights(), ms):
            if m is not None:
                assert torch.unique(w).numel() <= m.cluster_centers_.size
    xb, yb = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    with torch.no_grad():
        loss0 = float(t._get_error(xb, yb))
    acc = t.fine_tune_centroids(data, test, epochs=2, learning_rate=1e-3)
    assert len(acc) == 2 and all(0.0 <= a <= 1.0 for a in acc)
    with torch.no_grad():
        loss1 = float(t._get_error(xb, yb))
    assert loss1 < loss0, (loss0, loss1)
    moved = False
    for layer, ms in t.quantized_models_by_layer.items():
        for w, w0, m in zip(layer.get_weights(), before[layer], ms):
            if m is None:
                assert torch.equal(w, w0)          # tensors that passed through unquantized stay as they were
                continue
            cen = torch.from_numpy(m.cluster_centers_.ravel()).cuda()
            assert torch.equal(w.reshape(-1), cen[m.labels_device().long()])   # same indices, updated centroids
            moved |= not torch.equal(w, w0)
    assert moved
