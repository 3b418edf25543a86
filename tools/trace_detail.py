#!/usr/bin/env python3
"""Per-phase workgroup timeline of k_assign<accumulate> (start / prologue / loop / epilogue), per XCD."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat
from neural_network_compression_amd import kmeans, ops, pipeline, synth  # noqa: E402

which_iters = [int(a) for a in sys.argv[1:]] or [5, 20]
pruned = os.environ.get("PRUNE", "1") == "1"
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.weights((25_000_000,), 4000)).to(dev)
if pruned:
    ops.prune_(x, 1.0, True)
cdfs = pipeline.weight_distribution(x, pruned)
space = pipeline.initial_centroids(x, 8, "density", cdfs)
km = kmeans.DeviceKMeans(x, space)
L = nat.load()
tr = torch.zeros(4 * 1024 + 16, dtype=torch.int64, device=dev)
pct = lambda a: " ".join(f"{np.percentile(a, q):6.2f}" for q in (0, 10, 50, 90, 100))
for it in range(1, max(which_iters) + 1):
    if it in which_iters:
        tr.zero_()
        nat.check(L.nnc_debug_set_trace(tr.data_ptr()))
    km.iterate(1)
    st = km.status()
    if it in which_iters:
        torch.cuda.synchronize()
        nat.check(L.nnc_debug_set_trace(0))
        t4 = tr.cpu().numpy()[: 4 * 1024].reshape(-1, 4)
        live = t4[:, 0] > 0
        idx = np.nonzero(live)[0]
        t4 = t4[live]
        rel = (t4 - t4[:, 0].min()) * 0.01
        print(f"iter {it}: {len(t4)} wgs   (percentiles 0/10/50/90/100, us)")
        print("  start    ", pct(rel[:, 0]))
        print("  prologue ", pct(rel[:, 1] - rel[:, 0]))
        print("  loop     ", pct(rel[:, 2] - rel[:, 1]))
        print("  epilogue ", pct(rel[:, 3] - rel[:, 2]))
        print("  loop end ", pct(rel[:, 2]))
        print("  end      ", pct(rel[:, 3]))
        f = (tr.cpu().numpy()[4 * 1024: 4 * 1024 + 13] - tr.cpu().numpy()[4 * 1024]) * 0.01
        order = [(0, "start"), (1, "shards->partials"), (2, "empties+average"), (3, "shift+tol+state"), (8, "still-sorted test"), (9, "rank sort"),
                 (10, "distinct"), (4, "tables written"), (11, "zones raw"), (5, "zone scans"), (12, "cell binary search"), (7, "cells")]
        print("  still_sorted flag:", int(tr.cpu().numpy()[4 * 1024 + 13]))
        print("  k_finalize stamps (us since start): " + ", ".join(f"{nm} {f[i]:.2f}" for i, nm in order))
        for xcd in range(8):
            m = (idx % 8) == xcd
            print(f"   xcd {xcd}: start med {np.median(rel[m,0]):5.2f} loop med {np.median(rel[m,2]-rel[m,1]):5.2f} loop-end med/max {np.median(rel[m,2]):5.2f}/{rel[m,2].max():5.2f}")
    if st.paused:
        km._relocate_and_resume(st)
        st = km.status()
