#!/usr/bin/env python3
"""Diagnostics build: phase stamps of the LAST k_finalize call of the bench fit that settled an empty-cluster event in place."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat
from neural_network_compression_amd import kmeans, ops, pipeline, synth
dev = torch.device("cuda:0")
L = nat.load()
n = 25_000_000
x = torch.from_numpy(synth.weights((n,), 4000)).to(dev)
ops.prune_(x, 1.0, True)
cdfs = pipeline.weight_distribution(x, True)
space = pipeline.initial_centroids(x, 8, "density", cdfs)
tr = torch.zeros(8192, dtype=torch.int64, device=dev)
km = kmeans.DeviceKMeans(x, space, two_launch=True)
nat.check(L.nnc_debug_set_trace(tr.data_ptr()))
m, _ = km.fit()
torch.cuda.synchronize()
nat.check(L.nnc_debug_set_trace(0))
t = tr.cpu().numpy()[6 * 1024 + 20: 6 * 1024 + 27]
f = (t - t[0]) * 0.01
print("fit:", m.n_iter_, "iterations,", m.n_relocations_, "events")
print("last in-place event, us since the start of its k_finalize: selection starts %.1f, selection done %.1f, [average+shift done %.1f], order done %.1f, zones done %.1f, end %.1f" % (f[1], f[2], f[3], f[4], f[5], f[6]))
r = tr.cpu().numpy()[6 * 1024 + 30: 6 * 1024 + 38]
g = (r - t[0]) * 0.01
print("   inside the selection (us since the start of the k_finalize): enter %.1f, certain ends %.1f, stretch samples %.1f, barrier %.1f, keys in registers %.1f, rounds %.1f, vote + empties %.1f, edits %.1f" % tuple(g))
