"""The Lloyd loop inside one resident workgroup (csrc/nnc_lloyd.hpp, k_lloyd) against the launch-per-iteration form
(k_bounds + k_finalize, NNC_KM_TWO_LAUNCH) and against the oracle (mode B): same trajectory, bit for bit -- n_iter_, centres,
indices, decoded values, relocation events -- on plain, pruned, few-valued and crowded inputs, at every workgroup size the
launcher picks (K <= 128: 256 threads; above: 1024), whole fits and iteration by iteration.

Reference path: KMeans.fit as called from neural_network_compression/common/utility.py:237-238
(sklearn/cluster/_kmeans.py:624-752)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from neural_network_compression_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def km():
    assert torch.cuda.is_available()
    from neural_network_compression_amd import _native, build as _b, kmeans
    _b.build_native()
    _native.load()
    return kmeans


def _pruned(n, seed, q=1.0):
    x = synth.weights((n,), seed)
    orc.prune_weigth(x, q, True)
    return x


def _init(x, k, mode, seed=0):
    if mode == "linear":
        return np.linspace(x.min(), x.max(), k).astype(np.float32)
    if mode == "forgy":
        rng = np.random.RandomState(seed)
        return x[rng.randint(0, x.size, k)].astype(np.float32)
    if mode == "quantile":
        return np.quantile(x.astype(np.float64), np.linspace(0.0005, 0.9995, k)).astype(np.float32)
    raise ValueError(mode)


def _both(km, x, init, oracle=True):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    a_km = km.DeviceKMeans(t, init, loop=True)
    assert a_km.lloyd, "the one-workgroup loop must be the path taken"
    a, av = a_km.fit()
    b, bv = km.DeviceKMeans(t, init, two_launch=True).fit()
    assert a.n_iter_ == b.n_iter_, (a.n_iter_, b.n_iter_)
    assert a.stop_reason_ == b.stop_reason_
    assert np.array_equal(a.cluster_centers_.view(np.uint32), b.cluster_centers_.view(np.uint32))
    assert np.array_equal(a.labels_, b.labels_)
    assert torch.equal(av, bv)
    assert a.n_relocations_ == b.n_relocations_, (a.n_relocations_, b.n_relocations_)
    assert np.array_equal(a.counts_host_, b.counts_host_)
    if oracle:
        ob = orc.kmeans_lloyd(x, init, accum="B")
        assert a.n_iter_ == ob.n_iter_, (a.n_iter_, ob.n_iter_)
        assert np.array_equal(a.cluster_centers_.ravel(), ob.cluster_centers_.ravel())
        assert np.array_equal(a.labels_, ob.labels_)
    return a


@pytest.mark.parametrize("n,k,mode", [
    (600, 4, "linear"), (5_000, 16, "linear"), (70_001, 33, "linear"), (400_000, 64, "linear"), (400_000, 65, "quantile"),
    (300_000, 128, "quantile"), (300_000, 129, "quantile"), (300_000, 257, "quantile"), (200_000, 513, "quantile"),
    (150_000, 1025, "quantile"), (1_000_003, 256, "linear"),
])
def test_fit_equals_launch_per_iteration_and_oracle(km, n, k, mode):
    x = synth.weights((n,), 31 + n + k)
    _both(km, x, _init(x, k, mode))


@pytest.mark.parametrize("n,k,mode,seed", [
    (235_200, 16, "linear", 0), (235_200, 32, "forgy", 1), (235_200, 32, "forgy", 2), (30_000, 17, "forgy", 3),
    (627_200, 33, "forgy", 4), (500_000, 256, "forgy", 5), (500_000, 257, "forgy", 6), (120_000, 64, "forgy", 7),
])
def test_pruned_vectors_with_relocations(km, n, k, mode, seed):
    """sigma-pruned tensors (a plateau of zeros in the sorted vector) with inits that leave clusters empty: the loop pauses,
    the relocation chain settles the event, the loop goes on; crowded centres right after a relocation take the wide pass."""
    x = _pruned(n, 900 + seed)
    m = _both(km, x, _init(x, k, mode, seed))
    if mode == "forgy":
        assert m.n_relocations_ >= 0


def test_few_valued_and_constant_data(km):
    xe = np.full(5000, -0.375, dtype=np.float32)
    _both(km, xe, np.array([-0.375, 0.1, 0.2, -0.375], dtype=np.float32))
    x2 = np.where(np.arange(6000) % 3 == 0, np.float32(0.25), np.float32(-0.5)).astype(np.float32)
    _both(km, x2, np.array([-0.5, 0.0, 0.25, 0.3], dtype=np.float32))
    x3 = (np.arange(40_000) % 7).astype(np.float32) * np.float32(0.125)
    _both(km, x3, np.linspace(0, 0.75, 16).astype(np.float32))


def test_centres_float32_cannot_tell_apart(km):
    """Two centres one ulp apart make a whole neighbourhood undecided (a long stretch): the loop hands the iteration to the wide
    pass; three crowded centres likewise."""
    x = synth.weights((150_000,), 4242)
    init = np.linspace(x.min(), x.max(), 16).astype(np.float32)
    init[6] = np.nextafter(init[5], np.float32(1.0))
    _both(km, x, init)
    init2 = np.linspace(x.min(), x.max(), 16).astype(np.float32)
    init2[6] = np.nextafter(init2[5], np.float32(1.0))
    init2[7] = np.nextafter(init2[6], np.float32(1.0))
    _both(km, x, init2)


def test_huge_range_and_offsets(km):
    x = synth.weights((100_000,), 5, scale=1e-3)
    x[::997] *= 3e3
    _both(km, x, np.linspace(x.min(), x.max(), 32).astype(np.float32))
    y = synth.weights((100_000,), 6, scale=1e-4) + np.float32(7.5)
    _both(km, y, np.linspace(y.min(), y.max(), 16).astype(np.float32))


@pytest.mark.parametrize("n,k", [(50_000, 16), (200_000, 257)])
def test_iteration_by_iteration(km, n, k):
    """nnc_kmeans_iterate(1) on both forms: the centres after every single iteration are the same bits."""
    x = _pruned(n, 77 + k)
    init = _init(x, k, "quantile" if k > 64 else "linear")
    t = torch.from_numpy(x).cuda()
    a, b = km.DeviceKMeans(t, init, loop=True), km.DeviceKMeans(t, init, two_launch=True)
    for it in range(12):
        sa, sb = a.iterate_and_look(1), b.iterate_and_look(1)
        assert (sa.iter, sa.done, sa.paused, sa.n_empty) == (sb.iter, sb.done, sb.paused, sb.n_empty), it
        assert np.array_equal(a.centers(centred=True).view(np.uint32), b.centers(centred=True).view(np.uint32)), it
        if sa.done:
            break
        if sa.paused:
            a._relocate_and_resume(sa)
            b._relocate_and_resume(sb)


def test_iterate_runs_exactly_the_iterations_asked_for(km):
    x = synth.weights((300_000,), 5151)
    init = _init(x, 64, "linear")
    t = torch.from_numpy(x).cuda()
    a = km.DeviceKMeans(t, init, loop=True)
    for want in (1, 3, 8):
        before = a.status().iter
        st = a.iterate_and_look(want)
        assert st.paused or st.done or st.iter == before + want, (before, want, st.iter)
        if st.paused or st.done:
            break


def test_the_iterations_really_run_inside_the_loop(km):
    """The device's own counters: a plain fit runs (nearly) all of its iterations in the one-workgroup loop, none by the wide pair --
    by default up to NNC_KM_LOOP_KMAX centres, with loop=True beyond."""
    x = _pruned(400_000, 4711)
    t = torch.from_numpy(x).cuda()
    for k, kw in ((16, {}), (64, {}), (257, {"loop": True})):
        d = km.DeviceKMeans(t, _init(x, k, "quantile" if k > 64 else "linear"), **kw)
        assert d.lloyd
        m, _ = d.fit()
        st = d.loop_stats()
        assert st["loop_iterations"] >= m.n_iter_ and st["wide_iterations"] <= 1, (k, m.n_iter_, st)
    d = km.DeviceKMeans(t, _init(x, 257, "quantile"))
    assert not d.lloyd   # one compute unit's instruction rate is the bound there: launch per iteration
    d.fit()
    assert d.loop_stats()["loop_iterations"] == 0


def test_empty_clusters_are_settled_inside_the_loop(km):
    """Fits that pause for empty clusters (duplicate forgy draws / duplicate density centres on a pruned vector): under fit() the
    loop relocates the far samples itself (kl_relocate: candidates from the exact ranks of the iteration, proof that nothing behind
    them reaches the cut), falls back to the relocation chain where it cannot -- and the trajectory is the launch-per-iteration one,
    event for event."""
    in_loop = events = 0
    for n, k, seed in [(235_200, 32, 1), (235_200, 32, 2), (627_200, 33, 4), (500_000, 257, 6), (120_000, 64, 7), (300_000, 129, 8), (90_000, 24, 9)]:
        x = _pruned(n, 900 + seed)
        init = _init(x, k, "forgy", seed)
        t = torch.from_numpy(x).cuda()
        a_km = km.DeviceKMeans(t, init, loop=True)
        a, av = a_km.fit()
        b, bv = km.DeviceKMeans(t, init, two_launch=True).fit()
        assert (a.n_iter_, a.n_relocations_, a.stop_reason_) == (b.n_iter_, b.n_relocations_, b.stop_reason_), (n, k, seed)
        assert (a.reloc_tie_, a.n_reloc_multi_) == (b.reloc_tie_, b.n_reloc_multi_), (n, k, seed)
        assert np.array_equal(a.cluster_centers_.view(np.uint32), b.cluster_centers_.view(np.uint32))
        assert np.array_equal(a.labels_, b.labels_) and torch.equal(av, bv)
        ob = orc.kmeans_lloyd(x, init, accum="B")
        assert a.n_iter_ == ob.n_iter_ and np.array_equal(a.cluster_centers_.ravel(), ob.cluster_centers_.ravel())
        st = a_km.loop_stats()
        in_loop += st["relocated_in_loop"]
        events += a.n_relocations_
        assert st["relocated_in_loop"] <= a.n_relocations_
    assert events > 0 and in_loop > 0, (events, in_loop)


def test_finalize_step_settles_small_events_itself(km):
    """The launch-per-iteration form under fit(): the finalize step relocates the far samples of a small empty-cluster event itself
    (km_finalize_relocate: the selection of the resident loop, fed from the ranks k_bounds left) -- same trajectory as the oracle, and
    as the loop; stepping with iterate_and_look still shows every pause."""
    for n, k, seed in [(235_200, 32, 1), (627_200, 33, 4), (500_000, 257, 6), (300_000, 300, 8), (120_000, 64, 7)]:
        x = _pruned(n, 900 + seed)
        init = _init(x, k, "forgy", seed)
        t = torch.from_numpy(x).cuda()
        b, bv = km.DeviceKMeans(t, init, two_launch=True).fit()
        ob = orc.kmeans_lloyd(x, init, accum="B")
        assert b.n_iter_ == ob.n_iter_ and np.array_equal(b.cluster_centers_.ravel(), ob.cluster_centers_.ravel()), (n, k, seed)
        assert np.array_equal(b.labels_, ob.labels_)
        assert b.n_relocations_ == ob.reloc_info_.get("reloc_events", 0), (b.n_relocations_, ob.reloc_info_)
        # stepping: the pauses are the host's to see
        d = km.DeviceKMeans(t, init, two_launch=True)
        seen = 0
        for _ in range(60):
            st = d.iterate_and_look(1)
            if st.done:
                break
            if st.paused:
                seen += 1
                d._relocate_and_resume(st)
        assert seen == b.n_relocations_, (seen, b.n_relocations_)


def test_mass_events_settled_by_the_finalize_step_same_trajectory(km):
    """NNC_KM_MASS_IN_PLACE (opt-in, experiment of round 4): the finalize step of a launch-per-iteration pass settles MASS empty-cluster
    events itself (kl_relocate_mass: shallow windows at every end, deep ones only where the proof asks, undecided samples by a bound
    first) instead of the relocation chain.  Density / forgy inits with hundreds of duplicate centres on pruned data: the fit is the
    default one and the oracle's, bit for bit, event for event -- and the events really were settled in the launch."""
    import ctypes

    settled = events = 0
    for n, k, mode, seed in [(2_000_000, 257, "density", 1), (900_000, 257, "forgy", 2), (600_000, 300, "forgy", 3), (1_500_000, 129, "density", 4),
                             (400_000, 1025, "density", 5)]:
        x = _pruned(n, 700 + seed)
        if mode == "density":
            flat = x[x != 0]
            init = np.asarray(orc.init_space(x, {129: 7, 257: 8, 1025: 10}[k], "density", orc.get_weight_distribution(flat)), dtype=np.float32)
        else:
            init = _init(x, k, "forgy", seed)
        t = torch.from_numpy(x).cuda()
        a_km = km.DeviceKMeans(t, init, two_launch=True, mass_in_place=True)
        a, av = a_km.fit()
        b, bv = km.DeviceKMeans(t, init, two_launch=True).fit()
        assert (a.n_iter_, a.n_relocations_, a.stop_reason_) == (b.n_iter_, b.n_relocations_, b.stop_reason_), (n, k, mode)
        assert (a.reloc_tie_, a.n_reloc_multi_) == (b.reloc_tie_, b.n_reloc_multi_), (n, k, mode)
        assert np.array_equal(a.cluster_centers_.view(np.uint32), b.cluster_centers_.view(np.uint32)), (n, k, mode)
        assert np.array_equal(a.labels_, b.labels_) and torch.equal(av, bv)
        ob = orc.kmeans_lloyd(x, init, accum="B")
        assert a.n_iter_ == ob.n_iter_ and np.array_equal(a.cluster_centers_.ravel(), ob.cluster_centers_.ravel()), (n, k, mode)
        st = a_km.loop_stats()
        settled += st["relocated_in_loop"]
        events += a.n_relocations_
        assert a.n_reloc_multi_ >= 1, (n, k, mode)      # there were events of several clusters
    assert events >= 10 and settled >= events // 2, (events, settled)
