"""Worker of tests/test_gpu_multirank.py: one rank of a sharded compress_layer on GPU 0 (all ranks share the
card; collectives over gloo, which moves CUDA tensors through the host).  Prints one JSON line."""
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neural_network_compression_amd import pipeline, sharding, synth  # noqa: E402


def main():
    n_total, seed, q, bits, mode = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    group = None
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        group = dist.group.WORLD
    lo, hi = sharding.shard_bounds(n_total, world, rank)
    w = synth.weights((n_total,), seed)
    if mode == "forgy":
        np.random.seed(1234 + 17 * rank)   # only rank 0's global RNG state decides the draw (it is broadcast)
    x = torch.from_numpy(w[lo:hi].copy()).cuda()
    res = pipeline.compress_layer(x, q=q if q >= 0 else None, bits=bits, mode=mode, group=group, huffman=True)
    labels = res.model.labels_
    out = {
        "rank": rank, "lo": lo, "hi": hi, "n_iter": res.model.n_iter_, "stop": res.model.stop_reason_,
        "relocations": res.model.n_relocations_, "windowed": res.model.n_reloc_windowed_,
        "centers": hashlib.sha256(res.model.cluster_centers_.tobytes()).hexdigest(),
        "labels": hashlib.sha256(labels.astype(np.int32).tobytes()).hexdigest(),
        "values": hashlib.sha256(res.values.cpu().numpy().tobytes()).hexdigest(),
        "mask": hashlib.sha256(res.mask.cpu().numpy().tobytes()).hexdigest() if res.mask is not None else None,
        "counts": [int(c) for c in res.counts], "total_bits": int(res.total_bits),
        "nzeroed": res.nzeroed, "sigma": res.sigma,
    }
    if world == 1:
        # per-shard hashes of the unsharded result, for the comparison with 2 and 3 ranks
        out["shards"] = {}
        for ws_ in (2, 3):
            for r in range(ws_):
                a, b = sharding.shard_bounds(n_total, ws_, r)
                out["shards"][f"{ws_}:{r}"] = {
                    "labels": hashlib.sha256(labels[a:b].astype(np.int32).tobytes()).hexdigest(),
                    "values": hashlib.sha256(res.values[a:b].cpu().numpy().tobytes()).hexdigest(),
                    "mask": hashlib.sha256(res.mask[a:b].cpu().numpy().tobytes()).hexdigest() if res.mask is not None else None,
                }
    print("RESULT " + json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
