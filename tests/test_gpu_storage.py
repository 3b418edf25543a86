"""Compressed on-disk form (Deep Compression's Huffman stage, which the reference names but never wrote): the
GPU bit packer / unpacker against an independent bit-by-bit host construction, and a save -> load round trip that gives back
cluster_centers_[labels_] (neural_network_compression/common/utility.py:239) bit for bit.  Run with -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from neural_network_compression_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    from neural_network_compression_amd import _native, ops, pipeline, storage

    _native.load()
    return ops, pipeline, storage


def _host_stream(labels, lengths):
    """Canonical Huffman, MSB first, built bit by bit on the host (independent of the library's code table)."""
    k = lengths.size
    order = sorted((int(l), s) for s, l in enumerate(lengths) if l)
    codes, code, prev = {}, 0, order[0][0] if order else 0
    for l, s in order:
        code <<= (l - prev)
        codes[s] = (code, l)
        code += 1
        prev = l
    bits = []
    for s in labels:
        c, l = codes[int(s)]
        bits.extend((c >> (l - 1 - i)) & 1 for i in range(l))
    pad = (-len(bits)) % 32
    b = np.array(bits + [0] * pad, dtype=np.uint8).reshape(-1, 32)
    words = (b.astype(np.uint64) << np.arange(31, -1, -1, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    return words, len(bits)


@pytest.mark.parametrize("n,k,style", [(1, 1, "flat"), (5, 2, "flat"), (1000, 4, "skew"), (1024, 16, "flat"), (1025, 16, "skew"),
                                       (5000, 33, "one"), (40_000, 257, "skew"), (70_001, 1025, "flat"), (300_000, 5, "pruned")])
def test_pack_unpack_against_host_construction(mods, n, k, style):
    ops, _, storage = mods
    rng = np.random.RandomState(n + k)
    if style == "flat":
        lab = rng.randint(0, k, size=n)
    elif style == "skew":
        p = np.exp(-np.arange(k) * (12.0 / k)); p /= p.sum()
        lab = rng.choice(k, size=n, p=p)
    elif style == "one":
        lab = np.full(n, k - 2)
    else:  # a pruned layer: most indices are the zero cluster's
        lab = np.where(rng.rand(n) < 0.85, 2, rng.randint(0, k, size=n))
    dt = np.uint8 if k <= 256 else np.int16
    lab_d = torch.from_numpy(lab.astype(dt)).cuda()
    words, chunk_bits, lengths, total_bits = storage.encode_indices(lab_d, k)
    counts = np.bincount(lab, minlength=k)
    assert total_bits == int((counts * lengths.astype(np.int64)).sum())
    assert int(chunk_bits.astype(np.int64).sum()) == total_bits and chunk_bits.size == (n + 1023) // 1024
    if n <= 70_001:
        want, nbits = _host_stream(lab, lengths)
        assert nbits == total_bits
        assert np.array_equal(words.cpu().numpy().view(np.uint32), want)
    back = storage.decode_indices(words, chunk_bits, n, lengths, k, 1 if k <= 256 else 2)
    assert np.array_equal(back.cpu().numpy().astype(np.int64) & 0xFFFF, lab)
    # a flipped bit is noticed (wrong symbol count / length in some chunk) or at least changes the indices
    if total_bits > 64 and k > 2 and style != "one":
        broken = words.clone()
        broken[0] ^= 0x40000000
        try:
            b2 = storage.decode_indices(broken, chunk_bits, n, lengths, k, 1 if k <= 256 else 2)
            assert not np.array_equal(b2.cpu().numpy().astype(np.int64) & 0xFFFF, lab)
        except ValueError:
            pass


def test_save_and_load_a_compressed_network(mods, tmp_path):
    ops, pipeline, storage = mods
    tensors, want = {}, {}
    total = 0
    for li, (name, wshape, bshape) in enumerate(synth.LENET_300_100):
        for kind, shape, seed, q in (("w", wshape, 2000 + 2 * li, 1.0), ("b", bshape, 2001 + 2 * li, 0.1)):
            w = torch.from_numpy(synth.weights(shape, seed)).cuda()
            res = pipeline.compress_layer(w, q=q, bits=4, mode="linear")
            key = f"{name}.{kind}"
            if res.model is None:                       # too short for 16 centroids: stored raw
                tensors[key] = (shape, None, w)
                want[key] = w.reshape(shape)
            else:
                tensors[key] = (shape, res.model, None)
                want[key] = res.values.reshape(shape)
            total += int(np.prod(shape))
    path = str(tmp_path / "lenet300.nnc")
    size = storage.save_compressed(path, tensors)
    got = storage.load_compressed(path)
    assert set(got) == set(want)
    for key in want:
        assert got[key].shape == want[key].shape and torch.equal(got[key], want[key]), key
    bits_per_weight = 8.0 * size / total
    assert bits_per_weight < 2.5, bits_per_weight          # 32 bits -> about 2 (pruned at 1 sigma, 4-bit codebook, entropy coded)
    print(f"LeNet-300-100: {total} weights -> {size} bytes = {bits_per_weight:.2f} bits / weight ({32.0 / bits_per_weight:.1f}x)")


# ------------------------------------------------------------------ relative-index sparse form (Deep Compression section 3)
def _host_sparse_entries(lab, zero, dbits):
    """The entries position by position in plain Python: (distance - 1, index) per stored position, filler entries
    (distance 2^dbits, index = zero) for longer gaps, distances restarting at every chunk of 1024 positions."""
    D = 1 << dbits
    deltas, syms, per_chunk = [], [], []
    for base in range(0, len(lab), 1024):
        prev, cnt = base - 1, 0
        for i in range(base, min(base + 1024, len(lab))):
            if lab[i] == zero:
                continue
            gap = i - prev
            while gap > D:
                deltas.append(D - 1); syms.append(zero); gap -= D; cnt += 1
            deltas.append(gap - 1); syms.append(int(lab[i])); cnt += 1
            prev = i
        per_chunk.append(cnt)
    return np.array(deltas, dtype=np.int64), np.array(syms, dtype=np.int64), np.array(per_chunk, dtype=np.int64)


@pytest.mark.parametrize("n,k,density,dbits", [(1, 4, 1.0, 4), (7, 4, 0.0, 4), (1023, 16, 0.3, 4), (1024, 16, 0.05, 4), (1025, 16, 0.01, 4),
                                               (5000, 16, 0.001, 4), (40_000, 257, 0.1, 8), (40_000, 257, 0.002, 8), (70_001, 33, 0.5, 4),
                                               (3000, 16, 0.0, 8), (66_000, 5, 0.02, 1)])
def test_sparse_entries_against_host_construction(mods, n, k, density, dbits):
    ops, _, storage = mods
    rng = np.random.RandomState(n + k + dbits)
    zero = k // 2
    lab = np.where(rng.rand(n) < density, rng.randint(0, k, size=n), zero)
    dt = np.uint8 if k <= 256 else np.int16
    lab_d = torch.from_numpy(lab.astype(dt)).cuda()
    delta, sym, per_chunk = storage.encode_sparse(lab_d, zero, dbits)
    wd, ws, wc = _host_sparse_entries(lab, zero, dbits)
    assert np.array_equal(per_chunk.astype(np.int64), wc)
    assert np.array_equal(delta.cpu().numpy().astype(np.int64), wd)
    assert np.array_equal(sym.cpu().numpy().astype(np.int64) & 0xFFFF, ws)
    back = storage.decode_sparse(delta, sym, per_chunk, n, zero)
    assert np.array_equal(back.cpu().numpy().astype(np.int64) & 0xFFFF, lab)
    if delta.numel() * (1 << dbits) > 2048:     # entries that run past their chunk are noticed
        broken = torch.full_like(delta, (1 << dbits) - 1)
        table = np.zeros_like(per_chunk)
        table[0] = min(delta.numel(), 60000)
        with pytest.raises(ValueError):
            storage.decode_sparse(broken[: int(table[0])], sym[: int(table[0])], table, n, zero)


@pytest.mark.parametrize("sparsity,expect", [(0.0, "dense"), (0.68, None), (0.97, "sparse")])
def test_the_smaller_form_is_stored_and_decodes(mods, tmp_path, sparsity, expect):
    """pack_indices(form="auto") keeps the smallest of dense / sparse4 / sparse8 in bytes; each form on its own round-trips; a
    heavily pruned tensor is smaller in the relative-index form, an unpruned one in the dense stream."""
    ops, _, storage = mods
    import struct

    n, k = 300_000, 16
    rng = np.random.RandomState(int(sparsity * 100))
    lab = np.where(rng.rand(n) < sparsity, 7, rng.randint(0, k, size=n)).astype(np.uint8)
    lab_d = torch.from_numpy(lab).cuda()
    sizes = {}
    for form in ("dense", "sparse4", "sparse8"):
        body, bits, chosen = storage.pack_indices(lab_d, k, form=form)
        assert chosen == form
        back, pos = storage.unpack_indices(body, 0, k, n, 1, lab_d.device)
        assert pos == len(body) and np.array_equal(back.cpu().numpy(), lab)
        sizes[form] = len(body)
    body, bits, chosen = storage.pack_indices(lab_d, k, form="auto")
    assert len(body) == min(sizes.values()) or (sparsity < storage.SPARSE_MIN_ZERO_SHARE and chosen == "dense")
    if expect is not None:
        assert chosen.startswith(expect), (chosen, sizes)
    print(f"sparsity {sparsity}: " + ", ".join(f"{f} {8.0 * s / n:.3f} b/w" for f, s in sizes.items()) + f" -> {chosen}")


def test_store_report_writes_the_stored_network_and_its_ratio(mods, tmp_path):
    """Trainer.store_report after quantize: report.txt carries bits per weight and the compression ratio (Deep Compression's
    headline figure, which the reference's report.txt -- zero counts only, common/trainer.py:154-175 -- cannot give), and
    weights.nnc decodes to the network's quantized tensors bit for bit."""
    ops, pipeline, storage = mods
    from neural_network_compression_amd.common.trainer import LeNetDataset
    from neural_network_compression_amd.le_net_300_100_trainer import LeNet300100Trainer

    torch.manual_seed(0)
    tr = LeNet300100Trainer()
    tr.neural_network.cuda()
    for li, (layer_name, layer) in enumerate(tr.neural_network.get_config().items()):
        w, b = layer.get_weights()
        layer.set_weights([torch.from_numpy(synth.weights(tuple(w.shape), 2000 + 2 * li)).cuda(), torch.from_numpy(synth.weights(tuple(b.shape), 2001 + 2 * li)).cuda()])
    tr._prune_parameters(True)
    data = LeNetDataset(np.zeros((8, 784), dtype=np.float32), np.zeros((8,), dtype=np.int64))
    tr.quantize(data, False, 4, "linear")
    tr.store_report(str(tmp_path / "rep"))
    text = open(tmp_path / "rep" / "report.txt").read()
    assert "zeroed weights:" in text and "compression ratio" in text and "bits per weight" in text
    total = tr.compression_report["total"]
    assert total["n"] == 266_610 and 10.0 < total["compression_ratio"] < 40.0, total
    got = storage.load_compressed(str(tmp_path / "rep" / "weights.nnc"))
    for layer_name, layer in tr.neural_network.get_config().items():
        for kind, t in zip(("weights", "biases"), layer.get_weights()):
            assert torch.equal(got[f"{layer_name}.{kind}"], t), (layer_name, kind)
