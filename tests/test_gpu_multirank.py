"""Sharded fits with REAL shards: 2 and 3 processes share GPU 0 (collectives over gloo) and must reproduce the
single-process result bit for bit: same centres, iteration count, relocations, index histogram, and each rank's
slice of the labels / values / mask.  (The 8-GPU run itself is the driver's; this covers the cross-rank logic:
chunk-sum all-gather, min/max, per-iteration all-reduce of the sums, relocation key exchange.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "helpers", "multirank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, args):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, WORKER] + [str(a) for a in args], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se[-2000:]
        line = [l for l in so.splitlines() if l.startswith("RESULT ")][-1]
        outs.append(json.loads(line[7:]))
    return sorted(outs, key=lambda o: o["rank"])


@pytest.mark.parametrize("args", [
    (600_000, 71, 1.0, 5, "density"),     # pruned, duplicate initial centres: relocations
    (300_011, 72, -1, 4, "linear"),       # unpruned, ragged length
    (500_000, 73, 0.5, 5, "forgy"),
])
def test_sharded_fit_equals_single_process(args):
    assert torch.cuda.is_available()
    one = _run(1, args)[0]
    if args[4] == "density":
        assert one["relocations"] >= 1
    for world in (2, 3):
        many = _run(world, args)
        for o in many:
            assert o["n_iter"] == one["n_iter"] and o["stop"] == one["stop"], (world, o["n_iter"], one["n_iter"])
            assert o["centers"] == one["centers"] and o["relocations"] == one["relocations"]
            if args[4] == "density":
                assert o["windowed"] >= 1   # the sharded form of the windowed relocation ran
            assert o["counts"] == one["counts"] and o["total_bits"] == one["total_bits"]
            assert o["nzeroed"] == one["nzeroed"] and o["sigma"] == one["sigma"]
            ref = one["shards"][f"{world}:{o['rank']}"]
            assert o["labels"] == ref["labels"] and o["values"] == ref["values"] and o["mask"] == ref["mask"]


LAYERS_WORKER = os.path.join(HERE, "helpers", "layers_worker.py")


def _run_layers(world, shard_above):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, LAYERS_WORKER, str(shard_above)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
    return sorted(outs, key=lambda o: o["rank"])


@pytest.mark.parametrize("shard_above", [0, 2_000_000])
def test_layers_dealt_out_on_the_gpu(shard_above):
    """BASELINE configs[4]'s multi-GPU form rehearsed with real kernels: 2 and 3 processes share GPU 0 and deal twelve tensors of a
    layer list out between them (pipeline.compress_layers(group=...): longest first; shard_above: the longest ones sharded through the
    sharded fit instead).  Every rank ends up with the records of all tensors, and they are the single-process ones: n_iter_, centres,
    index histogram, Huffman total, sigma, zero count -- and the decoded tensor on the rank that owns it."""
    assert torch.cuda.is_available()
    one = _run_layers(1, 0)[0]["recs"]
    for world in (2, 3):
        many = _run_layers(world, shard_above)
        owner = many[0]["owner"]
        assert (min(owner) == -1) == (shard_above > 0)
        for o in many:
            assert o["owner"] == owner and [r["index"] for r in o["recs"]] == list(range(len(one)))
            for r, ref in zip(o["recs"], one):
                assert r["rank"] == owner[r["index"]]
                for key in ("n", "n_iter", "nzeroed", "sigma", "centers", "counts", "total_bits"):
                    assert r[key] == ref[key], (world, o["rank"], r["index"], key)
                assert r["has_result"] == (r["rank"] in (o["rank"], -1))
                if "values" in r:
                    assert r["values"] == ref["values"], (world, r["index"])
        # every tensor's device-side result lives on exactly one rank (or on all of them as shards)
        for i in range(len(one)):
            holders = [o["rank"] for o in many if o["recs"][i]["has_result"]]
            assert holders == ([owner[i]] if owner[i] >= 0 else list(range(world))), (i, holders)
