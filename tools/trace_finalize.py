#!/usr/bin/env python3
"""Phase stamps of k_finalize (thread 0) for a small-K and the headline fit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat
from neural_network_compression_amd import kmeans, ops, pipeline, synth

dev = torch.device("cuda:0")
L = nat.load()
order = [(0, "start"), (1, "shards->partials"), (2, "empties+average"), (3, "shift+tol+state"), (12, "perm filled"), (8, "still-sorted"), (9, "rank sort"),
         (10, "distinct"), (4, "tables"), (13, "pair zones"), (14, "barrier"), (15, "own zone + stores"), (11, "zones raw"), (5, "zone scans"), (7, "end")]
for n, bits, mode in ((2_359_296, 4, "linear"), (25_000_000, 8, "density")):
    x = torch.from_numpy(synth.weights((n,), 4000)).to(dev)
    ops.prune_(x, 1.0, True)
    cdfs = pipeline.weight_distribution(x, True) if mode == "density" else None
    space = pipeline.initial_centroids(x, bits, mode, cdfs)
    km = kmeans.DeviceKMeans(x, space, two_launch=True)
    tr = torch.zeros(8192, dtype=torch.int64, device=dev)
    acc = []
    late = []
    for it in range(1, 60):
        tr.zero_()
        if it >= 6:
            nat.check(L.nnc_debug_set_trace(tr.data_ptr()))
        km.iterate(1)
        st = km.status()
        torch.cuda.synchronize()
        nat.check(L.nnc_debug_set_trace(0))
        if it >= 6 and not st.paused:
            t = tr.cpu().numpy()[6 * 1024: 6 * 1024 + 16]
            if t[7] > t[0]:
                (late if it >= 22 else acc).append((t - t[0]) * 0.01)
        if st.done:
            break
        if st.paused:
            km._relocate_and_resume(st)
    f = np.median(np.array(acc), axis=0)
    print(f"n={n} K={2**bits}: k_finalize stamps, iterations 6-21, median of {len(acc)} (us since start): " + ", ".join(f"{nm} {f[i]:.2f}" for i, nm in order))
    if late:
        f = np.median(np.array(late), axis=0)
        print(f"n={n} K={2**bits}: k_finalize stamps, iterations 22-, median of {len(late)} (us since start): " + ", ".join(f"{nm} {f[i]:.2f}" for i, nm in order))
