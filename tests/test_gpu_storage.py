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
