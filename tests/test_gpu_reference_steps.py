"""Single Lloyd steps of the HIP path against the REFERENCE's own per-step outputs (run with -m gpu).

The fixtures (tests/golden/ref_goldens.*: ``estep/*``, ``step/*``, ``trace/*``) were produced by the very Cython routine
the reference's ``KMeans.fit`` loops over (sklearn ``lloyd_iter_chunked_dense``, reached from
neural_network_compression/common/utility.py:237-238), one thread.  Everything here goes through the low-level C ABI
(nnc_kmeans_init / set_centers / accumulate / finalize / partials / get_centers / assign) and is compared with the
reference's arrays directly -- the oracle's order-independent mode is not involved:

  * E-step labels and label counts: EXACT against the reference's label vectors (or their SHA-256);
  * per-cluster fixed-point sums: EXACT against sum(rint(x * 2^S)) over the reference's members (float64 on the host);
  * new centres: against the float64 mean of the reference's members to 1e-6 of the data scale (the reference itself,
    summing in float32, sits up to ~1e-4 away from that mean) and against the reference's centres to its summation error;
  * empty-cluster relocation: the relocated centres are single samples, compared bit for bit.
"""
import hashlib
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from neural_network_compression_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (only for the pruned input tensor of the traces)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


@pytest.fixture(scope="module")
def nnc():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from neural_network_compression_amd import _native, kmeans, ops

    _native.load()
    return SimpleNamespace(kmeans=kmeans, ops=ops, native=_native, dev=torch.device("cuda:0"))


def dev(nnc, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(nnc.dev)


def raw_km(nnc, x, c):
    """A fit whose data and centres are used AS THEY ARE (x_mean = 0), the way the step fixtures were made."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    nzv = x[x != 0]
    st = SimpleNamespace(mean=np.float32(0), var=np.float32(np.var(x)), min=np.float32(x.min()), max=np.float32(x.max()),
                         min_nonzero=np.float32(nzv.min()) if nzv.size else np.float32(np.inf),
                         max_nonzero=np.float32(nzv.max()) if nzv.size else np.float32(-np.inf),
                         n_negative=int((x < 0).sum()), n_zero=int((x == 0).sum()), n=x.size)
    return nnc.kmeans.DeviceKMeans(dev(nnc, x), np.asarray(c, dtype=np.float32), stats=st)


def labels_of(km, which=0):
    lab, _, _ = km.assign(which=which, labels=True)
    return lab.to(torch.int32).cpu().numpy() & 0xFFFF


def fix_sums(x64, labels, k, S):
    """sum over the members of every cluster of rint(x * 2^S) (exact in float64: |x * 2^S| < 2^29)."""
    q = np.rint(np.ldexp(x64, S)).astype(np.int64)
    out = np.zeros(k, dtype=np.int64)
    np.add.at(out, labels, q)
    return out


# ------------------------------------------------------------------ E-step: the reference's label vectors
def test_estep_kats_against_reference_labels(nnc, gold):
    keys = gold.keys("estep/")
    assert len(keys) >= 7
    for key in keys:
        c = gold.cases[key]
        x, cen = gold.arr(c["x"]), gold.arr(c["c"])
        km = raw_km(nnc, x, cen)
        got = labels_of(km)
        want = gold.arr(c["labels"]).astype(np.int32)
        assert np.array_equal(got, want), (key, int((got != want).sum()))


# ------------------------------------------------------------------ one full iteration: E + M (+ relocation)
@pytest.mark.parametrize("name", ["plain16", "one_empty", "three_empty", "pruned_gap", "all_equal"])
def test_single_step_against_reference(nnc, gold, name):
    c = gold.cases[f"step/{name}"]
    x, cen = gold.arr(c["x"]), gold.arr(c["c"])
    k = cen.size
    ref_labels = gold.arr(c["labels"]).astype(np.int64)
    ref_new = gold.arr(c["centers_new"])
    ref_wic = gold.arr(c["weight_in_clusters"])
    km = raw_km(nnc, x, cen)
    assert np.array_equal(labels_of(km), ref_labels), name
    # --- accumulate: the sums / counts the all-reduce would carry
    km.iterate(1)
    st = km.status()
    part = km.partials.cpu().numpy().copy()
    counts = np.bincount(ref_labels, minlength=k)
    n_empty = int((counts == 0).sum())
    assert n_empty == c["n_empty"]
    x64 = x.astype(np.float64)
    if n_empty:
        assert int(st.paused) == 1 and int(st.n_empty) == n_empty, name
        assert np.array_equal(part[k:], counts), name                                   # label counts: exact
        assert np.array_equal(part[:k], fix_sums(x64, ref_labels, k, km.fix_shift)), name  # fixed-point sums: exact
        km._relocate_and_resume(st)
        st = km.status()
        part = km.partials.cpu().numpy().copy()
    assert int(st.paused) == 0 and int(st.iter) == 1, name
    new = km.centers(which=0, centred=True)
    scale = float(np.abs(x).max())
    if name == "all_equal":
        # every distance is zero: nothing is relocated, the empty clusters copy the biggest one (scikit-learn's
        # _average_centers copies it "as it stands": its average for later indices, its raw sum for earlier ones)
        assert np.array_equal(new, ref_new), (new, ref_new)
        return
    # counts after relocation = the reference's weight_in_clusters
    assert np.array_equal(part[k:], ref_wic.astype(np.int64)), name
    was_empty = counts == 0
    # relocated clusters hold ONE far sample each: their centre is that sample, bit for bit the reference's choice
    assert np.array_equal(new[was_empty], ref_new[was_empty]), (name, new[was_empty], ref_new[was_empty])
    # the moved samples (their value is the relocated centre; their old cluster is where the reference had them)
    members = [x64[ref_labels == j] for j in range(k)]
    if n_empty:
        for j in np.nonzero(was_empty)[0]:
            v = np.float64(ref_new[j])
            (idx,) = np.nonzero(x64 == v)
            olds = {int(ref_labels[i]) for i in idx}
            assert len(olds) == 1, "ambiguous fixture"
            old = olds.pop()
            m = members[old]
            (pos,) = np.nonzero(m == v)
            members[old] = np.delete(m, pos[0])
            members[j] = np.array([v])
    for j in range(k):
        assert members[j].size == int(ref_wic[j]), (name, j)
        mean64 = members[j].mean()
        assert abs(float(new[j]) - mean64) <= 1e-6 * scale, (name, j, float(new[j]), mean64)   # the float64 mean of the reference's members
        assert abs(float(new[j]) - float(ref_new[j])) <= 1e-4 * scale, (name, j)              # the reference's float32 running sums
    shift = np.abs(new - cen)
    assert np.allclose(shift, gold.arr(c["shift"]), rtol=0, atol=1e-4 * scale), name


# ------------------------------------------------------------------ a whole recorded trajectory, one iteration at a time
@pytest.mark.parametrize("tkey", ["trace/cfg1.density2", "trace/cfg2.linear4"])
def test_every_iteration_of_a_reference_fit(nnc, gold, tkey):
    """BASELINE configs[0] / configs[1] on fc1 (784 x 300, pruned at 1 sigma): each of the reference's iterations is
    replayed from the REFERENCE's centres of the previous iteration (so no drift accumulates): labels by SHA-256 and
    label counts exact, the new centres against the float64 mean of those members and the reference's own centres."""
    from tests.golden.make_goldens import lenet300_tensors

    c = gold.cases[tkey]
    name, shape, seed = lenet300_tensors()[0]
    w = synth.weights(shape, seed)
    orc.prune_weigth(w, 1, True)
    x = w.ravel()
    x64 = x.astype(np.float64)
    init = gold.arr(c["init"])
    k = init.size
    km = nnc.kmeans.DeviceKMeans(dev(nnc, x), init)
    assert bits(km.x_mean) == c["x_mean_bits"] and bits(km.tol_) == c["tol_bits"]
    gc, gn, gs = gold.arr(c["centers_centred"]), gold.arr(c["counts"]), gold.arr(c["shift_tot"])
    mean = np.float32(km.x_mean)
    xc64 = (x - mean).astype(np.float32).astype(np.float64)    # the centred float32 data, as scikit-learn holds it
    scale = float(np.abs(xc64).max())
    L, ws, p = km.L, km.ws.data_ptr(), km.p
    import ctypes

    ties = 0
    for i in range(c["n_iter"]):
        if i > 0:
            prev = dev(nnc, gc[i - 1])
            nnc.native.check(L.nnc_kmeans_set_centers(ws, ctypes.byref(p), prev.data_ptr(), 1, km.stream))
        lab = labels_of(km)
        assert sha(lab.astype(np.int32)) == c["labels_sha"][i], (tkey, i)          # every centroid index of this E-step
        km.iterate(1)
        st = km.status()
        part = km.partials.cpu().numpy().copy()
        assert np.array_equal(part[k:], gn[i]), (tkey, i)                          # label counts
        assert np.array_equal(part[:k], fix_sums(xc64, lab, k, km.fix_shift)), (tkey, i)
        relocated = np.zeros(k, dtype=bool)
        if int(st.paused):
            relocated = gn[i] == 0
            before = int(st.reloc_ties)
            km._relocate_and_resume(st)
            st = km.status()
            assert int(st.paused) == 0
            if int(st.reloc_ties) > before:
                ties += 1
                continue   # two samples tie at the cut: which one scikit-learn keeps is numpy's introselect's business
        new = km.centers(which=0, centred=True)
        old_c = gc[i - 1] if i > 0 else (init - mean).astype(np.float32)
        # clusters the relocation touched: the receivers hold one sample (bit-exact), the donors lost one
        assert np.array_equal(new[relocated], gc[i][relocated]), (tkey, i)
        cnt = np.bincount(lab, minlength=k).astype(np.float64)
        s64 = np.bincount(lab, weights=xc64, minlength=k)
        for j in range(k):
            if relocated[j] or cnt[j] == 0:
                continue
            assert abs(float(new[j]) - float(gc[i][j])) <= 1e-4 * scale, (tkey, i, j)
            if not relocated.any():
                assert abs(float(new[j]) - s64[j] / cnt[j]) <= 1e-6 * scale, (tkey, i, j)
        if not relocated.any():
            e = 1e-4 * scale
            bound = 2.0 * np.sqrt(float(gs[i]) * k) * e + k * e * e
            assert abs(float(st.shift_tot) - float(gs[i])) <= bound + 1e-12, (tkey, i, float(st.shift_tot), float(gs[i]))
        assert int(st.iter) == i + 1
    # the one tie of these two trajectories is the one that sends configs[1]'s fc1 fit to another optimum
    assert ties == (1 if tkey.endswith("linear4") else 0)
