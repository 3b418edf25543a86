#!/usr/bin/env python3
"""Diagnostics build: phase stamps of the RESUMED k_finalize (after an empty-cluster relocation) on the bench fit, per event."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat
from neural_network_compression_amd import kmeans, ops, pipeline, synth

dev = torch.device("cuda:0")
L = nat.load()
order = [(0, "start"), (1, "sums in"), (2, "empties+average"), (3, "shift+tol+state"), (8, "still-sorted"), (9, "rank sort"),
         (10, "distinct"), (4, "tables"), (13, "pair zones"), (14, "barrier"), (15, "own zone + stores"), (11, "zones raw"), (5, "zone scans"), (7, "end")]
n = 25_000_000
x = torch.from_numpy(synth.weights((n,), 4000)).to(dev)
ops.prune_(x, 1.0, True)
cdfs = pipeline.weight_distribution(x, True)
space = pipeline.initial_centroids(x, 8, "density", cdfs)
km = kmeans.DeviceKMeans(x, space, two_launch=True)
tr = torch.zeros(8192, dtype=torch.int64, device=dev)
for it in range(1, 30):
    km.iterate(1)
    st = km.status()
    if st.done:
        break
    if st.paused:
        ne = int(st.n_empty)
        tr.zero_()
        torch.cuda.synchronize()
        nat.check(L.nnc_debug_set_trace(tr.data_ptr()))
        km._relocate_and_resume(st)
        torch.cuda.synchronize()
        nat.check(L.nnc_debug_set_trace(0))
        t = tr.cpu().numpy()[6 * 1024: 6 * 1024 + 16]
        f = (t - t[0]) * 0.01
        print(f"event at iteration {it - 1} ({ne:3d} empty): " + ", ".join(f"{nm} {f[i]:.1f}" for i, nm in order), flush=True)
