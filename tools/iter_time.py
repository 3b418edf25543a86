#!/usr/bin/env python3
"""Per-iteration cost of the Lloyd loop on the bench vector (25 M weights, K = 257 density init), rank-boundary form against
the streaming form: wall time of a batch of iterations enqueued by one call, no look-ins in between."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_network_compression_amd import _native as nat, kmeans, pipeline, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = nat.load()
w = synth.weights((n,), 4000)
x = torch.from_numpy(w).cuda()
from neural_network_compression_amd import ops
ops.prune_(x, 1.0, True)
st = kmeans.LayerStats(x)
xs = kmeans.sorted_copy(x, st)
cdfs = pipeline.weight_distribution_sorted(xs, st)
space = pipeline.initial_centroids(x, bits, "density", cdfs)
for rb in (True, False):
    for rep in range(2):
        km = kmeans.DeviceKMeans(x, space, stats=st, x_sorted=xs, rank_boundaries=rb)
        # get past the relocations of the first iterations
        for _ in range(40):
            s = km.iterate_and_look(1)
            if s.paused:
                km._relocate_and_resume(s)
            if s.done or (int(s.iter) >= 12 and not s.paused):
                break
        torch.cuda.synchronize()
        it0 = int(km.status().iter)
        t0 = time.perf_counter()
        km.iterate(10)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s = km.status()
        print(f"rank_boundaries={rb} rep={rep}: {int(s.iter) - it0} iterations in {dt * 1e6:.0f} us -> {dt * 1e6 / max(1, int(s.iter) - it0):.1f} us / iteration (done={s.done} paused={s.paused})")
    model, _ = kmeans.DeviceKMeans(x, space, stats=st, x_sorted=xs, rank_boundaries=rb).fit()
    print(f"  full fit: n_iter={model.n_iter_} relocations={model.n_relocations_} centres sha={hash(model.cluster_centers_.tobytes()) & 0xffffffff:08x}")
