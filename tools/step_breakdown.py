#!/usr/bin/env python3
"""Where one bench.py step spends its wall time (host-synchronised section timings)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, ops, pipeline, synth  # noqa: E402

n = 25_000_000
dev = torch.device("cuda:0")
w0 = torch.from_numpy(synth.weights((n,), 4000)).to(dev)
pipeline.compress_layer(w0.clone(), q=1.0, bits=8, mode="density")  # warm-up


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(2):
    t = [sync()]
    x = w0.clone(); t.append(sync())
    mask, stats, nz = ops.prune_(x, 1.0, True); t.append(sync())
    cdfs = pipeline.weight_distribution(x, True); t.append(sync())
    space = pipeline.initial_centroids(x, 8, "density", cdfs); t.append(sync())
    km = kmeans.DeviceKMeans(x, space); t.append(sync())
    # the fit loop, with relocation time separated
    reloc_t, reloc_n, status_n = 0.0, 0, 0
    orig = km._relocate_and_resume

    def timed(st):
        global reloc_t, reloc_n
        a = sync(); orig(st); reloc_t += sync() - a; reloc_n += 1

    km._relocate_and_resume = timed
    model, vals = km.fit(True); t.append(sync())
    counts = ops.bincount(model.labels_compact_, km.k).cpu().numpy(); t.append(sync())
    ops.huffman_lengths(counts); t.append(sync())
    names = ["clone", "prune", "cdf", "init space", "kmeans setup (moments+sort+init)", "fit (all)", "bincount", "huffman"]
    d = np.diff(t) * 1e3
    print(f"--- rep {rep}: total {1e3 * (t[-1] - t[0]):.2f} ms; n_iter {model.n_iter_}, relocations {reloc_n} ({reloc_t * 1e3:.2f} ms inside fit)")
    for nm, v in zip(names, d):
        print(f"   {nm:36s} {v:8.3f} ms")
