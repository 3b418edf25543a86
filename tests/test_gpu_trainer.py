"""GPU tests of the reference's call surface (Trainer / LeNet300100Trainer /
run_experiment_with_lenet300100) and of the per-layer pipeline, against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from neural_network_compression_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def trainer_mod():
    assert torch.cuda.is_available()
    from neural_network_compression_amd import _native, build as _b
    _b.build_native()  # no-op when csrc/libnnc_hip.so is up to date
    _native.load()
    from neural_network_compression_amd import le_net_300_100_trainer, main, pipeline
    from neural_network_compression_amd.common import trainer
    return le_net_300_100_trainer, trainer, main, pipeline


def _load_synth_weights(net):
    ws = {}
    for li, (name, wshape, bshape) in enumerate(synth.LENET_300_100):
        layer = getattr(net, name)
        w = synth.weights(wshape, 2000 + 2 * li)
        b = synth.weights(bshape, 2000 + 2 * li + 1)
        layer.set_weights([torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda()])
        ws[name] = (w, b)
    return ws


def test_prune_and_reset_parameters_match_oracle(trainer_mod):
    lt, tr, _, _ = trainer_mod
    tr.Trainer.pruned_indexes_by_layer.clear()
    t = lt.LeNet300100Trainer()
    ws = _load_synth_weights(t.neural_network)
    t._prune_parameters(True)
    q = {"dense1": (1, 0.1), "dense2": (1, 0.1), "out": (0.5, 0)}
    for name, (w, b) in ws.items():
        layer = getattr(t.neural_network, name)
        mw, mb = t.pruned_indexes_by_layer[layer]
        wo, bo = w.copy(), b.copy()
        omw = orc.prune_weigth(wo, q[name][0], True)
        omb = orc.prune_weigth(bo, q[name][1], True)
        assert np.array_equal(mw.cpu().numpy(), omw) and np.array_equal(mb.cpu().numpy(), omb)
        gw, gb = layer.get_weights()
        assert np.array_equal(gw.cpu().numpy(), wo) and np.array_equal(gb.cpu().numpy(), bo)
        # an optimiser step revives pruned weights; _reset_pruned_parameters zeroes them again
        gw.add_(0.5)
        gb.add_(0.5)
    t._reset_pruned_parameters()
    for name, (w, b) in ws.items():
        layer = getattr(t.neural_network, name)
        mw, mb = t.pruned_indexes_by_layer[layer]
        gw, gb = layer.get_weights()
        assert bool((gw[mw] == 0).all()) and bool((gb[mb] == 0).all())
        assert bool((gw[~mw] != 0).all())


@pytest.mark.parametrize("bits,mode,with_cdf", [(2, "density", True), (4, "linear", False)])
def test_quantize_matches_oracle_layer_by_layer(trainer_mod, bits, mode, with_cdf):
    lt, tr, _, _ = trainer_mod
    tr.Trainer.pruned_indexes_by_layer.clear()
    t = lt.LeNet300100Trainer()
    ws = _load_synth_weights(t.neural_network)
    t._prune_parameters(True)
    test = tr.LeNetDataset(np.random.RandomState(0).rand(64, 784).astype(np.float32), np.zeros(64, dtype=np.int64))
    acc = t.quantize(test, with_cdf, bits, mode)
    assert 0.0 <= acc <= 1.0
    q = {"dense1": (1, 0.1), "dense2": (1, 0.1), "out": (0.5, 0)}
    for name, (w, b) in ws.items():
        layer = getattr(t.neural_network, name)
        for got, ref, qq in zip(layer.get_weights(), (w.copy(), b.copy()), q[name]):
            orc.prune_weigth(ref, qq, True)
            cdfs = None
            if with_cdf:
                flat = ref.ravel()
                cdfs = orc.get_weight_distribution(flat[flat != 0])
            want, km = orc.get_quantized_weight(ref.copy(), bits=bits, mode=mode, cdfs=cdfs, accum="device")
            assert np.array_equal(got.cpu().numpy(), want), (name, ref.shape)
            if km is not None:
                assert len(np.unique(got.cpu().numpy())) <= km.cluster_centers_.size


@pytest.mark.parametrize("bits,mode,with_cdf", [(2, "density", True), (4, "linear", False), (5, "density", True), (5, "linear", False)])
def test_quantize_in_reference_arithmetic_is_the_reference(trainer_mod, gold, bits, mode, with_cdf):
    """BASELINE configs[0] and [1] through the TRAINER entry (Trainer.quantize, /root/reference/.../common/trainer.py:42-72) with
    arith="reference", reloc="reference": every tensor of LeNet-300-100 then holds exactly what the reference's
    get_quantized_weight returned for it (goldens made by the reference itself: the decoded tensor by SHA-256, the centres by
    value, n_iter_) -- the tie at a relocation cut that sends the default fit of dense1 (linear, 4 bits) to another optimum included.
    The default arithmetic is timed beside it."""
    import hashlib
    import time

    lt, tr, _, _ = trainer_mod
    test = tr.LeNetDataset(np.random.RandomState(0).rand(64, 784).astype(np.float32), np.zeros(64, dtype=np.int64))
    took = {}
    for arith, reloc in (("auto", "auto"), ("reference", "reference")):
        tr.Trainer.pruned_indexes_by_layer.clear()
        t = lt.LeNet300100Trainer()
        _load_synth_weights(t.neural_network)
        t._prune_parameters(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.quantize(test, with_cdf, bits, mode, arith=arith, reloc=reloc)
        torch.cuda.synchronize()
        took[arith] = time.perf_counter() - t0
    for name, _, _ in synth.LENET_300_100:
        layer = getattr(t.neural_network, name)
        for kind, got, model in zip("wb", layer.get_weights(), t.quantized_models_by_layer[layer]):
            c = gold.cases[f"quant/cfg2/l300.{name}.{kind}/{mode}{bits}"]
            if c["passthrough"]:
                assert model is None
                continue
            assert model.arith_ == "reference" and model.n_iter_ == c["n_iter"], (name, kind, model.n_iter_, c["n_iter"])
            assert np.array_equal(model.cluster_centers_.ravel(), gold.arr(c["centers"]).ravel()), (name, kind)
            assert hashlib.sha256(np.ascontiguousarray(got.cpu().numpy()).tobytes()).hexdigest() == c["quantized_sha256"], (name, kind)
    print(f"LeNet-300-100 {mode}{bits}: Trainer.quantize {took['auto'] * 1e3:.1f} ms (default arithmetic), {took['reference'] * 1e3:.1f} ms (reference arithmetic)")


def test_run_experiment_surface(trainer_mod, tmp_path, monkeypatch):
    _, tr, main, _ = trainer_mod
    tr.Trainer.pruned_indexes_by_layer.clear()
    monkeypatch.chdir(tmp_path)
    main.run_experiment_with_lenet300100(train_epochs=1, prune_train_epochs=1, semi_prune_train_epochs=1,
                                         maximum_centroid_bits=2, k_means_initialization_mode="density",
                                         with_cumulative_weight_distribution=True, experiment_name="smoke")
    d = tmp_path / "LeNet300100_smoke"
    rep = (d / "report.txt").read_text()
    assert "layer: dense1" in rep and "zeroed weights:" in rep
    acc = (d / "accuracies.txt").read_text()
    assert "after quantization" in acc


def test_compress_layer_pipeline_matches_oracle(trainer_mod):
    _, _, _, pipeline = trainer_mod
    w = synth.weights((768, 768), 5000)
    x = torch.from_numpy(w.copy()).cuda()
    res = pipeline.compress_layer(x, q=1, bits=4, mode="linear")
    wo = w.copy()
    omask = orc.prune_weigth(wo, 1, True)
    ob = orc.kmeans_lloyd(wo.ravel(), orc.init_space(wo, 4, "linear"), accum="B")
    assert np.array_equal(res.mask.cpu().numpy().astype(bool).ravel(), omask.ravel())
    assert res.nzeroed == int(omask.sum())
    assert res.model.n_iter_ == ob.n_iter_
    assert np.array_equal(res.model.cluster_centers_.ravel(), ob.cluster_centers_.ravel())
    assert np.array_equal(res.model.labels_, ob.labels_)
    assert np.array_equal(res.values.cpu().numpy(), ob.cluster_centers_.ravel()[ob.labels_])
    counts = np.bincount(ob.labels_, minlength=16)
    assert np.array_equal(res.counts, counts)
    ol, oh, ot = orc.huffman_lengths(counts)
    assert np.array_equal(res.code_lengths, ol) and res.total_bits == ot


def test_sharded_code_path_with_one_rank_group(trainer_mod):
    """The torch.distributed branch (RCCL all-reduce of the 2K int64 partials, chunk-sum
    all-gather, sharded CDF / init / relocation) on a 1-rank NCCL group: same results as the
    plain single-GPU path."""
    import torch.distributed as dist

    _, _, _, pipeline = trainer_mod
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    try:
        group = dist.group.WORLD
        w = synth.weights((300_000,), 7100)
        for bits, mode, q in [(4, "density", 1), (5, "forgy", 1), (4, "linear", 1)]:
            np.random.seed(11)
            a = pipeline.compress_layer(torch.from_numpy(w.copy()).cuda(), q=q, bits=bits, mode=mode)
            np.random.seed(11)
            b = pipeline.compress_layer(torch.from_numpy(w.copy()).cuda(), q=q, bits=bits, mode=mode, group=group)
            assert a.nzeroed == b.nzeroed and a.sigma == b.sigma
            assert torch.equal(a.mask, b.mask)
            assert a.model.n_iter_ == b.model.n_iter_ and a.model.n_relocations_ == b.model.n_relocations_
            assert np.array_equal(a.model.cluster_centers_, b.model.cluster_centers_)
            assert np.array_equal(a.model.labels_, b.model.labels_)
            assert torch.equal(a.values, b.values)
            assert np.array_equal(a.counts, b.counts) and a.total_bits == b.total_bits
        # the same with the exchange inside the C library: the library's own RCCL communicator (nnc_comm_*),
        # nnc_kmeans_iterate_sharded and nnc_kmeans_relocate_windowed_sharded (key merge on the device)
        from neural_network_compression_amd import sharding

        comm = sharding.RcclComm(group, torch.device("cuda:0"))
        try:
            assert comm.world == 1 and comm.rank == 0
            t = torch.tensor([5, -7, 11], dtype=torch.int64, device="cuda")
            assert comm.allreduce_(t.clone(), "sum").tolist() == [5, -7, 11] and comm.allreduce_(t.clone(), "max").tolist() == [5, -7, 11]
            for bits, mode, q in [(4, "density", 1), (5, "forgy", 1), (8, "density", None)]:
                np.random.seed(11)
                a = pipeline.compress_layer(torch.from_numpy(w.copy()).cuda(), q=q, bits=bits, mode=mode)
                np.random.seed(11)
                b = pipeline.compress_layer(torch.from_numpy(w.copy()).cuda(), q=q, bits=bits, mode=mode, group=group, comm=comm)
                assert a.model.n_iter_ == b.model.n_iter_ and a.model.n_relocations_ == b.model.n_relocations_
                assert a.model.n_reloc_windowed_ == b.model.n_reloc_windowed_ and a.model.reloc_tie_ == b.model.reloc_tie_
                assert np.array_equal(a.model.cluster_centers_, b.model.cluster_centers_)
                assert np.array_equal(a.model.labels_, b.model.labels_)
                assert torch.equal(a.values, b.values)
                assert np.array_equal(a.counts, b.counts) and a.total_bits == b.total_bits
            # the whole loop inside the library (nnc_kmeans_fit_sharded) through fits that pause for empty clusters: duplicate forgy
            # draws on a pruned vector -- windowed relocations with their collectives, the look-ins and the batch sizing
            events = 0
            for seed, bits in [(1, 5), (2, 5), (3, 8), (4, 8), (5, 6)]:
                np.random.seed(seed)
                a = pipeline.compress_layer(torch.from_numpy(w.copy()).cuda(), q=1, bits=bits, mode="forgy")
                np.random.seed(seed)
                b = pipeline.compress_layer(torch.from_numpy(w.copy()).cuda(), q=1, bits=bits, mode="forgy", group=group, comm=comm)
                assert (a.model.n_iter_, a.model.n_relocations_, a.model.stop_reason_) == (b.model.n_iter_, b.model.n_relocations_, b.model.stop_reason_), (seed, bits)
                assert np.array_equal(a.model.cluster_centers_.view(np.uint32), b.model.cluster_centers_.view(np.uint32))
                assert np.array_equal(a.model.labels_, b.model.labels_) and torch.equal(a.values, b.values)
                assert np.array_equal(a.counts, b.counts) and a.total_bits == b.total_bits
                events += b.model.n_relocations_
            assert events > 0
        finally:
            comm.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_merge_of_the_ranks_farthest_keys(trainer_mod):
    """The key merge of the sharded windowed relocation (nnc_merge_keys: every rank's descending list of farthest-sample
    keys -> the overall top, descending, padding dropped) against a plain sort, for 1..8 lists with ties between lists."""
    from neural_network_compression_amd import _native as nat

    L = nat.load()
    rng = np.random.RandomState(9)
    for nlists in (1, 2, 3, 8):
        for per in (2, 17, 258, 1033):
            lists = []
            for _ in range(nlists):
                nreal = rng.randint(0, per + 1)
                keys = rng.randint(1, 50 if per < 100 else 1 << 40, size=nreal).astype(np.int64)   # small range: ties between lists
                keys = np.sort(keys)[::-1]
                pad = np.full(per - nreal, rng.choice([0, -1]), dtype=np.int64)
                lists.append(np.concatenate([keys, pad]))
            flat = np.concatenate(lists)
            want = np.sort(flat[flat > 0])[::-1][:per]
            want = np.concatenate([want, np.zeros(per - want.size, dtype=np.int64)])
            out = torch.empty(per, dtype=torch.int64, device="cuda")
            nat.check(L.nnc_merge_keys(torch.from_numpy(flat).cuda().data_ptr(), nlists, per, out.data_ptr(), per,
                                       torch.cuda.current_stream().cuda_stream))
            assert np.array_equal(out.cpu().numpy(), want), (nlists, per)


def test_lenet5_trainer_prune_and_quantize_match_oracle(trainer_mod):
    """BASELINE configs[2]: LeNet-5 conv + dense tensors (Keras layouts), prune at (1, 0.1) sigma, 5-bit forgy
    k-means; the tensors too short for 32 centroids pass through.  Layer by layer against the oracle."""
    _, tr, main, _ = trainer_mod
    from neural_network_compression_amd import le_net_5_trainer
    tr.Trainer.pruned_indexes_by_layer.clear()
    t = le_net_5_trainer.LeNet5Trainer()
    ws = {}
    attr = {"conv1": "conv1", "conv2": "conv2", "dense1": "dense", "out": "logits"}   # synth's names -> the reference's
    for li, (sname, wshape, bshape) in enumerate(synth.LENET_5):
        name = attr[sname]
        layer = getattr(t.neural_network, name)
        w = synth.weights(wshape, 3000 + 2 * li)
        b = synth.weights(bshape, 3000 + 2 * li + 1)
        layer.set_weights([torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda()])
        ws[name] = (w, b)
    t._prune_parameters(True)
    q = {"conv1": (1, 0.1), "conv2": (1, 0.1), "dense": (1, 0.1), "logits": (0.5, 0)}
    for name, (w, b) in ws.items():
        mw, mb = t.pruned_indexes_by_layer[getattr(t.neural_network, name)]
        assert np.array_equal(mw.cpu().numpy(), orc.prune_weigth(w.copy(), q[name][0], True))
        assert np.array_equal(mb.cpu().numpy(), orc.prune_weigth(b.copy(), q[name][1], True))
    test = tr.LeNetDataset(np.random.RandomState(0).rand(16, 28, 28, 1).astype(np.float32), np.zeros(16, dtype=np.int64))
    np.random.seed(77)
    acc = t.quantize(test, False, 5, "forgy")
    assert 0.0 <= acc <= 1.0
    np.random.seed(77)   # the oracle draws the same forgy indices in the same tensor order
    for name, (w, b) in ws.items():
        layer = getattr(t.neural_network, name)
        for got, ref, qq in zip(layer.get_weights(), (w.copy(), b.copy()), q[name]):
            orc.prune_weigth(ref, qq, True)
            want, km = orc.get_quantized_weight(ref.copy(), bits=5, mode="forgy", cdfs=None, accum="device")
            assert np.array_equal(got.cpu().numpy(), want), (name, ref.shape)
            assert (km is None) == (ref.size < 33)


def test_run_experiment_lenet5_surface(trainer_mod, tmp_path, monkeypatch):
    _, tr, main, _ = trainer_mod
    tr.Trainer.pruned_indexes_by_layer.clear()
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(main, "_synthetic", lambda n_train=1024, n_test=256, _f=main._synthetic: _f(n_train, n_test))
    main.run_experiment_with_lenet5(train_epochs=1, prune_train_epochs=1, semi_prune_train_epochs=1,
                                    maximum_centroid_bits=3, k_means_initialization_mode="linear",
                                    with_cumulative_weight_distribution=False, experiment_name="smoke")
    rep = (tmp_path / "LeNet5_smoke" / "report.txt").read_text()
    assert "layer: conv1" in rep and "layer: dense" in rep and "layer: logits" in rep
