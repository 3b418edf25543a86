#!/usr/bin/env python3
"""Which iterations of the bench fit pause for an empty cluster (one iteration per look-in)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, ops, pipeline, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
x = torch.from_numpy(synth.weights((n,), 4000)).cuda()
ops.prune_(x, 1.0, True)
ls = kmeans.LayerStats(x, n, None)
xs = kmeans.sorted_copy(x, ls)
cdfs = pipeline.weight_distribution_sorted(xs, ls, None)
space = pipeline.initial_centroids(x, bits, "density", cdfs, None, n)
km = kmeans.DeviceKMeans(x, space, stats=ls, x_sorted=xs, n_total=n, n_min=n)
ev = []
while True:
    st = km.iterate_and_look(1)
    if st.done:
        break
    if st.paused:
        ev.append((int(st.iter), int(st.n_empty), int(st.paused)))
        km._relocate_and_resume(st)
print("iterations", int(st.iter), "events (iter, n_empty, paused):", ev)
