"""BASELINE configs[4] over the GPUs of a node, rehearsed on the CPU: two and three processes over gloo deal the tensors of a
layer list out with the product's own code (pipeline.partition_layers / compress_layers(group=...)), with the CPU oracle
standing in for the per-tensor HIP path (``_compress``).  What must hold: every tensor is worked on by exactly one rank, every
rank ends up with the records of ALL tensors, in order, and they are the single-process ones.
Reference loop being dealt out: /root/reference/neural_network_compression/common/trainer.py:50-70."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from neural_network_compression_amd import pipeline, synth
from oracle import oracle as orc

# a small model's worth of tensors: long, short, one too short for 16 centres ("not enough bits")
SHAPES = [(300, 100), (100,), (64, 48), (5000,), (10,), (2000,), (48, 48), (1500,), (700,), (333,)]


class _HostTensor:
    """numel() + the array: what compress_layers needs of a tensor when the per-tensor function is the oracle"""

    def __init__(self, a):
        self.a = a

    def numel(self):
        return int(self.a.size)


def _oracle_layer(t, q=None, bits=4, mode="linear", **_kw):
    w = t.a.copy()
    mask = orc.prune_weigth(w, q, True)
    if w.size < 2 ** bits + 1:
        return pipeline.LayerResult(mask, int(mask.sum()), None, None, w, None, None, None, None, None)
    km = orc.kmeans_lloyd(w.ravel(), orc.init_space(w, bits, mode), accum="B")
    km.stop_reason_, km.n_relocations_ = "tol", 0
    counts = np.bincount(km.labels_, minlength=2 ** bits).astype(np.int64)
    lengths, lhist, total = orc.huffman_lengths(counts)
    return pipeline.LayerResult(mask, int(mask.sum()), None, None, None, km, counts, lengths, lhist, int(total))


def _tensors():
    return [_HostTensor(synth.weights(s, 6100 + i)) for i, s in enumerate(SHAPES)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tensors = _tensors()
        sizes = [t.numel() for t in tensors]
        owner = pipeline.partition_layers(sizes, world)
        held = [t if owner[i] == rank else None for i, t in enumerate(tensors)]     # a rank holds only its own tensors
        recs = pipeline.compress_layers(held, group=dist.group.WORLD, sizes=sizes, _compress=_oracle_layer, q=1.0, bits=4, mode="linear")
        assert [r.index for r in recs] == list(range(len(sizes)))
        for i, r in enumerate(recs):
            assert r.rank == owner[i] and r.n == sizes[i]
            assert (r.result is not None) == (owner[i] == rank)      # device-side results stay where they were made
        for r in recs:
            r.result = None
        pickle.dump(recs, open(os.path.join(out_dir, f"recs{rank}.pkl"), "wb"))
        # a rank that lacks one of its own tensors must say so, not hang the gather: checked before any collective
        if rank == 0:
            bad = list(held)
            bad[owner.index(0)] = None
            with pytest.raises(ValueError):
                pipeline.compress_layers(bad, group=dist.group.WORLD, sizes=sizes, _compress=_oracle_layer, q=1.0, bits=4, mode="linear")
        with pytest.raises(ValueError):   # forgy draws from the global generator in layer order: refused on every rank alike
            pipeline.compress_layers(held, group=dist.group.WORLD, sizes=sizes, _compress=_oracle_layer, q=1.0, bits=4, mode="forgy")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_layers_dealt_out_to_ranks_equal_single_process(tmp_path, world):
    orc.build()
    single = [_oracle_layer(t, q=1.0, bits=4, mode="linear") for t in _tensors()]
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    per_rank = [pickle.load(open(tmp_path / f"recs{r}.pkl", "rb")) for r in range(world)]
    for recs in per_rank:
        assert len(recs) == len(single)
        for rec, one in zip(recs, single):
            assert rec.nzeroed == one.nzeroed
            if one.model is None:
                assert rec.centers is None and rec.n_iter == 0
                continue
            assert rec.n_iter == one.model.n_iter_
            assert rec.centers.tobytes() == one.model.cluster_centers_.ravel().tobytes()
            assert np.array_equal(rec.counts, one.counts) and np.array_equal(rec.code_lengths, one.code_lengths)
            assert rec.total_bits == one.total_bits
    # all ranks hold the same records
    for recs in per_rank[1:]:
        for a, b in zip(recs, per_rank[0]):
            assert a.rank == b.rank and a.n_iter == b.n_iter and a.total_bits == b.total_bits


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_partition_every_tensor_once_and_balanced(world):
    sizes = [int(np.prod(s)) for _, s in synth.gpt2_small_layers()]
    owner = pipeline.partition_layers(sizes, world)
    assert len(owner) == 122 and all(0 <= o < world for o in owner)
    load = [sum(pipeline.layer_cost(n) for n, o in zip(sizes, owner) if o == r) for r in range(world)]
    assert max(load) <= 1.1 * sum(load) / world
    assert owner == pipeline.partition_layers(sizes, world)                      # a pure function of the sizes
    # the token embedding sharded, everything else dealt out
    o2 = pipeline.partition_layers(sizes, world, shard_above=10_000_000)
    assert (o2[0] == -1) == (world > 1) and all(o >= 0 for o in o2[1:])
