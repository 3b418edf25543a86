"""GPU edge cases: tiny tensors, unaligned views, K > 256 (16-bit labels), K = 1025, constant
and near-constant tensors, huge dynamic range, all against the oracle (mode B) bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from neural_network_compression_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def km_mod():
    assert torch.cuda.is_available()
    from neural_network_compression_amd import _native, build as _b, kmeans, ops
    _b.build_native()  # no-op when csrc/libnnc_hip.so is up to date
    _native.load()
    return kmeans, ops


def _check(kmeans, x, init, **kw):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    model, vals = kmeans.DeviceKMeans(t, init, **kw).fit()
    ob = orc.kmeans_lloyd(x, init, accum="B")
    assert model.n_iter_ == ob.n_iter_, (model.n_iter_, ob.n_iter_)
    assert np.array_equal(model.cluster_centers_.ravel(), ob.cluster_centers_.ravel())
    assert np.array_equal(model.labels_, ob.labels_), int((model.labels_ != ob.labels_).sum())
    assert np.array_equal(vals.cpu().numpy(), ob.cluster_centers_.ravel()[ob.labels_])
    assert np.array_equal(model.counts_device_.cpu().numpy(), np.bincount(ob.labels_, minlength=len(init)))
    return model


@pytest.mark.parametrize("n,k", [(17, 16), (33, 32), (5, 4), (1000, 4), (4097, 16), (8193, 5), (70_001, 33)])
def test_small_and_ragged_sizes(km_mod, n, k):
    kmeans, _ = km_mod
    x = synth.weights((n,), 100 + n)
    _check(kmeans, x, np.linspace(x.min(), x.max(), k).astype(np.float32))


def test_unaligned_input_view(km_mod):
    kmeans, _ = km_mod
    x = synth.weights((50_003,), 9)
    t = torch.from_numpy(x).cuda()[3:]  # 4-byte aligned only: scalar kernels
    init = np.linspace(x[3:].min(), x[3:].max(), 16).astype(np.float32)
    model, vals = kmeans.DeviceKMeans(t, init, sort=False).fit()
    ob = orc.kmeans_lloyd(x[3:], init, accum="B")
    assert model.n_iter_ == ob.n_iter_ and np.array_equal(model.labels_, ob.labels_)
    assert np.array_equal(model.cluster_centers_.ravel(), ob.cluster_centers_.ravel())


@pytest.mark.parametrize("k", [257, 513, 1025])
def test_many_centroids_16bit_labels(km_mod, k):
    kmeans, _ = km_mod
    x = synth.weights((300_000,), 1000 + k)
    qs = np.quantile(x.astype(np.float64), np.linspace(0.0005, 0.9995, k)).astype(np.float32)
    model = _check(kmeans, x, qs)
    assert model.labels_compact_.dtype == torch.int16
    assert model.labels_.max() == k - 1 or model.labels_.max() < k


@pytest.mark.parametrize("n,k", [(400_000, 64), (700, 16), (5_000, 16), (70_001, 256)])
def test_sorted_and_unsorted_iterations_agree(km_mod, n, k):
    """Both forms of the fit (value-sorted copy with windowed relocation; the vector as it stands with the full-pass
    relocation) give the same model; which one a tensor gets is a matter of its length only."""
    kmeans, _ = km_mod
    x = synth.weights((n,), 77 + n)
    init = np.linspace(x.min(), x.max(), k).astype(np.float32)
    a, _ = kmeans.DeviceKMeans(torch.from_numpy(x).cuda(), init, sort=True).fit()
    b, _ = kmeans.DeviceKMeans(torch.from_numpy(x).cuda(), init, sort=False).fit()
    assert a.n_iter_ == b.n_iter_ and np.array_equal(a.labels_, b.labels_)
    assert np.array_equal(a.cluster_centers_, b.cluster_centers_)


def test_constant_and_two_valued_tensors(km_mod):
    kmeans, _ = km_mod
    xe = np.full(5000, -0.375, dtype=np.float32)
    _check(kmeans, xe, np.array([-0.375, 0.1, 0.2, -0.375], dtype=np.float32))
    x2 = np.where(np.arange(6000) % 3 == 0, np.float32(0.25), np.float32(-0.5)).astype(np.float32)
    _check(kmeans, x2, np.array([-0.5, 0.0, 0.25, 0.3], dtype=np.float32))


def test_huge_dynamic_range_and_offsets(km_mod):
    kmeans, _ = km_mod
    x = synth.weights((100_000,), 5, scale=1e-3)
    x[::997] *= 3e3                      # outliers four orders of magnitude above the bulk
    _check(kmeans, x, np.linspace(x.min(), x.max(), 32).astype(np.float32))
    y = synth.weights((100_000,), 6, scale=1e-4) + np.float32(7.5)   # large mean, tiny spread
    _check(kmeans, y, np.linspace(y.min(), y.max(), 16).astype(np.float32))
    z = synth.weights((100_000,), 7, scale=1e-30)                     # tiny magnitudes (big fixed-point shift)
    _check(kmeans, z, np.linspace(z.min(), z.max(), 16).astype(np.float32))


def test_prune_edge_cases(km_mod):
    _, ops = km_mod
    from neural_network_compression_amd.common import utility
    for n in (1, 2, 7, 8, 9, 4095, 4097):
        w = synth.weights((n,), 50 + n)
        wo = w.copy()
        omask = orc.prune_weigth(wo, 0.7, True)
        mask = utility.prune_weigth(w, 0.7, True)
        assert np.array_equal(mask, omask) and np.array_equal(w, wo), n
    e = np.zeros((0,), dtype=np.float32)
    assert utility.prune_weigth(e, 1, False).shape == (0,)
    w = synth.weights((3, 5, 7), 3)
    wo = w.copy()
    assert np.array_equal(utility.prune_weigth(w, 0, True), orc.prune_weigth(wo, 0, True))  # q = 0: nothing pruned
    assert not utility.prune_weigth(w, 0, True).any()


def _fit_pair(kmeans, x, init):
    """windowed ('auto') and full-pass relocation against the oracle; returns the 'auto' fit object"""
    ob = orc.kmeans_lloyd(x, init, accum="B")
    kms = {}
    for mode in ("auto", "full"):
        km = kmeans.DeviceKMeans(torch.from_numpy(np.ascontiguousarray(x)).cuda(), init, reloc=mode)
        model, _ = km.fit()
        assert model.n_iter_ == ob.n_iter_, (mode, model.n_iter_, ob.n_iter_)
        assert np.array_equal(model.cluster_centers_.ravel(), ob.cluster_centers_.ravel()), mode
        assert np.array_equal(model.labels_, ob.labels_), mode
        assert np.array_equal(model.counts_device_.cpu().numpy(), np.bincount(ob.labels_, minlength=len(init))), mode
        kms[mode] = km
    assert kms["full"].n_reloc_windowed == 0
    assert kms["auto"].n_relocations == kms["full"].n_relocations
    return kms["auto"]


def test_windowed_relocation_on_sorted_vectors(km_mod):
    """Empty clusters on vectors long enough to be iterated value-sorted: the candidates come from
    windows around the cluster boundaries, the device proves the choice or the full pass repeats it."""
    kmeans, _ = km_mod
    # centres in the pruned gap and beyond the tails
    x = synth.weights((300_000,), 31)
    x[np.abs(x) < 0.05] = 0
    km = _fit_pair(kmeans, x, np.linspace(x.min() * 1.5, x.max() * 1.5, 64).astype(np.float32))
    assert km.n_relocations >= 1 and km.n_reloc_windowed >= 1
    # duplicate initial centres (what the density init produces), many empties at once
    x = synth.weights((200_000,), 32)
    qs = np.quantile(x.astype(np.float64), np.linspace(0.01, 0.99, 40)).astype(np.float32)
    init = np.repeat(qs, 5)[:190]
    km = _fit_pair(kmeans, x, init)
    assert km.n_relocations >= 1 and km.n_reloc_windowed >= 1
    # few distinct values, long runs of equal samples around every boundary
    x = (np.round(synth.weights((150_000,), 33) * 200) / 200).astype(np.float32)
    init = np.repeat(np.linspace(x.min(), x.max(), 12).astype(np.float32), 2)
    km = _fit_pair(kmeans, x, init)
    assert km.n_relocations >= 1
    # a cluster narrower than the window, more empties than the smallest window
    x = np.concatenate([synth.weights((100_000,), 34), np.linspace(0.4, 0.5, 40).astype(np.float32)])
    init = np.concatenate([np.full(100, 0.45, dtype=np.float32), np.linspace(-0.2, 0.2, 28).astype(np.float32)])
    km = _fit_pair(kmeans, x, init)
    assert km.n_relocations >= 1


def test_windowed_relocation_full_size_equals_full_pass(km_mod):
    """BASELINE configs[3] workload (25 M weights pruned at 1 sigma, density init, K = 257): thirteen
    relocation events; the windowed selection and the full distance pass give the same fit, bit for bit."""
    kmeans, ops = km_mod
    from neural_network_compression_amd import pipeline
    x = torch.from_numpy(synth.weights((25_000_000,), 4000)).cuda()
    ops.prune_(x, 1.0, True)
    cdfs = pipeline.weight_distribution(x, True)
    space = pipeline.initial_centroids(x, 8, "density", cdfs)
    fits = {}
    for mode in ("auto", "full"):
        km = kmeans.DeviceKMeans(x, space, reloc=mode)
        model, vals = km.fit()
        fits[mode] = (km, model, vals)
    (ka, ma, va), (kf, mf, vf) = fits["auto"], fits["full"]
    assert ka.n_reloc_windowed >= 5 and kf.n_reloc_windowed == 0 and ka.n_relocations == kf.n_relocations
    assert ma.n_iter_ == mf.n_iter_ and ma.stop_reason_ == mf.stop_reason_
    assert np.array_equal(ma.cluster_centers_, mf.cluster_centers_)
    assert torch.equal(ma.labels_compact_, mf.labels_compact_)
    assert torch.equal(va, vf)


def test_pruned_sort_equals_full_sort(km_mod):
    """nnc_sort_pruned_f32 (zeros partitioned out, not sorted) gives the value order of a full sort."""
    import ctypes
    from neural_network_compression_amd import _native as nat
    kmeans, ops = km_mod
    L = nat.load()
    for n, thr in [(1_000_003, 0.05), (70_000, 0.2), (300_000, 0.0)]:
        x = synth.weights((n,), 77 + n)
        x[np.abs(x) < thr] = 0
        x[::1000] = -0.0
        if thr == 0.2:
            x[x > 0] = 0  # no positives at all
        t = torch.from_numpy(x).cuda()
        mm, signs = ops.minmax_signs(t)
        n_neg, n_zero = (int(v) for v in signs.cpu().numpy())
        assert n_neg == int((x < 0).sum()) and n_zero == int((x == 0).sum())
        assert np.array_equal(mm.cpu().numpy()[:2], np.array([x.min(), x.max()], dtype=np.float32))
        out = torch.empty_like(t)
        wsb = L.nnc_sort_pruned_workspace_bytes(n, n_neg, n_zero)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        nat.check(L.nnc_sort_pruned_f32(t.data_ptr(), n, n_neg, n_zero, out.data_ptr(), ws.data_ptr(), wsb, ops._stream(t)))
        assert np.array_equal(out.cpu().numpy(), np.sort(x))  # array_equal: -0.0 == +0.0


def test_weight_distribution_from_sorted_copy(km_mod):
    """The CDF of the non-zero weights from ranks in the sorted copy equals the histogram-kernel path
    (and through it the reference goldens), on pruned and unpruned vectors."""
    kmeans, ops = km_mod
    from neural_network_compression_amd import pipeline
    for n, thr in [(300_000, 0.05), (100_000, 0.0), (70_000, 0.11)]:
        x = synth.weights((n,), 500 + n)
        x[np.abs(x) < thr] = 0
        x[::997] = -0.0
        t = torch.from_numpy(x).cuda()
        st = kmeans.LayerStats(t)
        nzv = x[x != 0]
        assert st.min_nonzero == nzv.min() and st.max_nonzero == nzv.max()
        assert st.min == x.min() and st.max == x.max()
        xs = kmeans.sorted_copy(t, st)
        a = pipeline.weight_distribution_sorted(xs, st)
        b = pipeline.weight_distribution(t, skip_zeros=True)
        c = orc.get_weight_distribution(nzv)
        for u, v, w in zip(a, b, c):
            assert np.array_equal(u, v) and np.array_equal(u, w)


@pytest.mark.parametrize("k", [2, 16, 32, 33, 64, 65, 256])
@pytest.mark.parametrize("grid_log2", [0, 6, 10, 11, 12, 14, 15])
def test_cell_grid_size_does_not_change_the_fit(km_mod, k, grid_log2):
    """The E-step's cell grid is an accelerator, never an approximation: every size gives the oracle's fit, and so do
    both builders of the table -- k_finalize's own (up to 64 centres on grids of up to 2^11 cells, which is also
    what grid_log2 = 0, the default, selects there) and k_cells (the rest).  Crowded initial centres make cells
    with many candidates (the side list) on the coarse grids."""
    kmeans, _ = km_mod
    x = synth.weights((40_000,), 4242 + k)
    x[np.abs(x) < np.float32(0.8) * x.std()] = 0
    init = np.concatenate([np.linspace(x.min(), x.max(), k - k // 2), np.full(k // 2, np.float32(1e-3))]).astype(np.float32)
    init[k // 2:] += np.arange(k - k // 2, dtype=np.float32)[: k - k // 2][: init[k // 2:].size] * np.float32(1e-7)
    _check(kmeans, x, init, grid_log2=grid_log2)


@pytest.mark.parametrize("n", [1, 7, 768, 8192, 8193, 70_001, 1_000_003])
def test_layer_statistics_in_one_call(km_mod, n):
    """nnc_layer_stats_f32 (one enqueue) = numpy on the same float32 vector, bit for bit: mean, variance, min / max
    over all and over the non-zero weights, sign counts."""
    kmeans, _ = km_mod
    x = synth.weights((n,), 31 + n)
    if n > 4:
        x[np.abs(x) < np.float32(0.7) * x.std()] = 0
    st = kmeans.LayerStats(torch.from_numpy(x).cuda())
    assert np.float32(st.mean).tobytes() == np.mean(x).tobytes() and np.float32(st.var).tobytes() == np.var(x).tobytes()
    assert st.min == x.min() and st.max == x.max()
    nzv = x[x != 0]
    if nzv.size:
        assert st.min_nonzero == nzv.min() and st.max_nonzero == nzv.max()
    else:
        assert np.isinf(st.min_nonzero) and np.isinf(st.max_nonzero)
    assert st.n_negative == int((x < 0).sum()) and st.n_zero == int((x == 0).sum())


@pytest.mark.parametrize("n", [1, 3, 5, 1000, 8191, 8192, 8193, 70_001, 1_000_003])
def test_prune_with_statistics_in_one_pass(km_mod, n):
    """nnc_prune_stats_f32 = nnc_prune_f32 followed by nnc_minmax_signs_f32 of the pruned tensor, from one pass."""
    _, ops = km_mod
    w = synth.weights((n,), 4400 + n % 97)
    if n > 10:
        w[::7] = 0.0
        w[3] = -0.0
    for q, smooth in ((1.0, True), (0.0, True), (0.05, False), (100.0, True)):
        a = torch.from_numpy(w.copy()).cuda()
        b = torch.from_numpy(w.copy()).cuda()[0:]          # (aligned)
        m1, s1, z1 = ops.prune_(a, q, smooth)
        mm1, sg1 = ops.minmax_signs(a)
        m2, s2, z2, mm2, sg2 = ops.prune_stats_(b, q, smooth)
        assert torch.equal(a, b) and torch.equal(m1, m2) and torch.equal(s1, s2) and int(z1.item()) == int(z2.item()), (n, q)
        assert torch.equal(mm1, mm2) and torch.equal(sg1, sg2), (n, q, mm1.tolist(), mm2.tolist(), sg1.tolist(), sg2.tolist())
    if n > 8:   # a 4-byte-aligned view
        base = torch.from_numpy(np.concatenate([np.zeros(1, np.float32), w])).cuda()
        c, d = base.clone()[1:], base.clone()[1:]
        m1, s1, z1 = ops.prune_(c, 1.0, True)
        mm1, sg1 = ops.minmax_signs(c)
        m2, s2, z2, mm2, sg2 = ops.prune_stats_(d, 1.0, True)
        assert torch.equal(c, d) and torch.equal(m1, m2) and torch.equal(mm1, mm2) and torch.equal(sg1, sg2)


@pytest.mark.parametrize("n", [1, 7, 64, 4095, 4096, 4097, 70_001, 1_048_576, 3_000_017])
def test_bounded_sort_of_a_pruned_vector(km_mod, n):
    """nnc_sort_pruned_bounded_f32 (hand-written radix sort of compact keys) = a plain ascending sort."""
    import ctypes
    from neural_network_compression_amd import _native as nat

    _, ops = km_mod
    L = nat.load()
    rs = np.random.RandomState(n % 1000)
    for case in range(4):
        w = synth.weights((n,), 6600 + case + n % 89, scale=float([0.05, 3.0, 1e-3, 0.05][case]))
        if case == 2:
            w = np.abs(w)                                   # one sign only
        if case == 3 and n > 10:
            w[rs.randint(0, n, size=n // 3)] = w[0]         # many equal values
        x = torch.from_numpy(w.copy()).cuda()
        q = [1.0, 0.5, 1.5, 1.0][case]
        mask, stats, nz, mm, signs = ops.prune_stats_(x, q, True)
        thr = float(stats.cpu().numpy()[1])
        mmh, sg = mm.cpu().numpy(), signs.cpu().numpy()
        n_neg, n_zero = int(sg[0]), int(sg[1])
        bits = int(L.nnc_sort_pruned_bounded_bits(float(mmh[0]), float(mmh[1]), thr, n_neg, n - n_neg - n_zero))
        if n_neg + n_zero == n and n_zero == n:
            continue                                          # everything pruned: nothing to sort
        if not thr > 0:
            assert bits == 0                                  # (a single weight: sigma 0, no threshold, the form does not apply)
            continue
        assert 0 < bits <= 27, (n, case, bits, thr, mmh)
        out = torch.empty_like(x)
        wsb = int(L.nnc_sort_pruned_bounded_workspace_bytes(n - n_zero))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        nat.check(L.nnc_sort_pruned_bounded_f32(x.data_ptr(), n, n_neg, n_zero, float(mmh[0]), float(mmh[1]), thr, out.data_ptr(), ws.data_ptr(), wsb,
                                                torch.cuda.current_stream().cuda_stream))
        want = torch.sort(x).values
        assert torch.equal(out, want), (n, case, int((out != want).sum()))
    # the form does not apply: no threshold, or a range of more than 27 bits
    assert L.nnc_sort_pruned_bounded_bits(-1.0, 1.0, 0.0, 5, 5) == 0
    assert L.nnc_sort_pruned_bounded_bits(-1.0, 1.0, 1e-30, 5, 5) == 0
    assert L.nnc_sort_pruned_bounded_bits(-0.5, 0.5, 0.06, 5, 5) == 26


def test_bounded_sort_reports_weights_outside_its_bounds(km_mod):
    """nnc_sort_pruned_bounded_f32 clamps a weight outside the bounds it was given (or a NaN) instead of writing out of range, and
    says so in the flag word of its workspace; with honest bounds the flag stays 0.  nnc_compress_layer_f32 reads the flag and hands
    such a tensor (a NaN weight: the statistics pass ignores it) to the step-by-step path instead of fitting a sorted copy that is
    not the tensor's."""
    import ctypes
    from neural_network_compression_amd import _native as nat, pipeline

    _, ops = km_mod
    L = nat.load()
    n = 70_001
    w = synth.weights((n,), 8801)
    x = torch.from_numpy(w.copy()).cuda()
    mask, stats, nz, mm, signs = ops.prune_stats_(x, 1.0, True)
    thr = float(stats.cpu().numpy()[1])
    mmh, sg = mm.cpu().numpy(), signs.cpu().numpy()
    n_neg, n_zero = int(sg[0]), int(sg[1])
    stream = torch.cuda.current_stream().cuda_stream
    wsb = int(L.nnc_sort_pruned_bounded_workspace_bytes(n - n_zero))
    for vmax, want_flag in ((float(mmh[1]), 0), (float(mmh[1]) * 0.5, 1)):   # honest bounds; an upper bound that half the tail exceeds
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        out = torch.empty_like(x)
        nat.check(L.nnc_sort_pruned_bounded_f32(x.data_ptr(), n, n_neg, n_zero, float(mmh[0]), vmax, thr, out.data_ptr(), ws.data_ptr(), wsb, stream))
        torch.cuda.synchronize()
        addr = L.nnc_sort_pruned_bounded_flag(ws.data_ptr(), n - n_zero)
        off = addr - ws.data_ptr()
        flag = int(ws[off: off + 4].view(torch.int32).item())
        assert (flag != 0) == bool(want_flag), (vmax, flag)
        if not want_flag:
            assert torch.equal(out, torch.sort(x).values)
    # a NaN (or an infinite) weight: the statistics pass ignores a NaN for min / max, so the compact-key sort would clamp it -- the
    # layer call notices (flag, non-finite mean) and hands the tensor back; the fit then refuses it as KMeans.fit does
    # (sklearn's input validation: ValueError), on the one-call path and on the step-by-step path alike, long and short tensors
    for n_, bad in ((200_000, np.nan), (200_000, np.inf), (3000, np.nan)):
        wn = synth.weights((n_,), 8802)
        wn[n_ // 3] = bad
        for native in (True, False):
            with pytest.raises(ValueError, match="NaN or infinity"):
                pipeline.compress_layer(torch.from_numpy(wn.copy()).cuda(), q=1.0, bits=4, mode="linear", native=native)
    from neural_network_compression_amd.common import utility
    with pytest.raises(ValueError, match="NaN or infinity"):
        wn = synth.weights((50_000,), 8803)
        wn[7] = np.nan
        utility.get_quantized_weight(wn, bits=4, mode="linear")


@pytest.mark.parametrize("n", [1, 2, 63, 8191, 8192, 8193, 100_003, 2_000_000, 9_000_001])
def test_plain_sort_of_an_unpruned_vector(km_mod, n):
    """nnc_sort_f32 (hand-written radix sort: four passes of 8 bits over the ordered images of the floats, decoupled look-back;
    the one-time value sort of an UNPRUNED vector, BASELINE configs[3] as SURVEY 8(d) writes it) = a plain ascending sort, for
    Gaussian data, data with many equal values, huge dynamic range, negative zeros, infinities, and a misaligned view."""
    from neural_network_compression_amd import _native as nat

    _, ops = km_mod
    L = nat.load()
    rs = np.random.RandomState(n % 977)
    for case in range(5):
        w = synth.weights((n + 1,), 9100 + case + n % 97, scale=float([0.05, 1e-20, 3e4, 0.05, 1.0][case]))
        if case == 3 and n > 10:
            w[rs.randint(0, n, size=n // 2)] = w[1]             # many equal values
            w[rs.randint(0, n, size=n // 50 + 1)] = -0.0
            w[rs.randint(0, n, size=n // 50 + 1)] = 0.0
        if case == 4 and n > 10:
            w[rs.randint(0, n, size=5)] = np.inf
            w[rs.randint(0, n, size=5)] = -np.inf
            w[: n // 2] *= np.float32(1e-30)                      # denormals and tiny values next to order-one values
        full = torch.from_numpy(w).cuda()
        x = full[1:] if case == 2 else full[:n]                   # case 2: a view that is not 16-byte aligned
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        wsb = int(L.nnc_sort_workspace_bytes(n))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        nat.check(L.nnc_sort_f32(x.data_ptr(), n, out.data_ptr(), ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream))
        want = torch.sort(x).values
        assert torch.equal(out, want), (n, case, int((out != want).sum()))
        # the multiset of bit patterns is the input's (nothing invented, -0.0 and +0.0 both kept)
        if n <= 2_000_000:
            assert np.array_equal(np.sort(out.cpu().numpy().view(np.uint32)), np.sort(x.cpu().numpy().view(np.uint32))), (n, case)
