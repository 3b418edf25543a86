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
