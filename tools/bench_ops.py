#!/usr/bin/env python3
"""Per-operator timings on a resident 25 M vector (torch events around each call, median of reps)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, ops, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
dev = torch.device("cuda:0")
w = torch.from_numpy(synth.weights((n,), 4000)).to(dev)


def timeit(name, fn, reps=7, bytes_=None):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    t = float(np.median(ts))
    extra = f"  {bytes_ / t / 1e6:7.2f} TB/s" if bytes_ else ""
    print(f"{name:34s} {t:9.1f} us{extra}")


x = w.clone()
timeit("clone (d2d copy)", lambda: w.clone(), bytes_=8 * n)
timeit("chunk_sums", lambda: ops.chunk_sums(x), bytes_=4 * n)
timeit("moments (mean,var,std)", lambda: ops.moments(x), bytes_=8 * n)
thr = torch.tensor([0.05], device=dev)
timeit("threshold_mask_ (in place)", lambda: ops.threshold_mask_(x, thr), bytes_=9 * n)
x = w.clone()
timeit("prune_ (sigma + threshold)", lambda: ops.prune_(x, 1.0, True), bytes_=17 * n)
timeit("minmax", lambda: ops.minmax(x), bytes_=4 * n)
timeit("minmax skip zeros", lambda: ops.minmax(x, True), bytes_=4 * n)
steps = torch.linspace(-0.3, 0.3, 32, device=dev)
timeit("hist31", lambda: ops.hist31(x, steps, True), bytes_=4 * n)
mask = (x == 0)
timeit("apply_mask_", lambda: ops.apply_mask_(x, mask), bytes_=5 * n)
init = np.linspace(-0.25, 0.25, 256).astype(np.float32)
km = kmeans.DeviceKMeans(x, init)
timeit("DeviceKMeans.__init__ (moments+sort)", lambda: kmeans.DeviceKMeans(x, init), reps=3)
timeit("sorted copy", lambda: kmeans.sorted_copy(x, km.stats), bytes_=4 * n)
timeit("assign labels+values", lambda: km.assign(0, True, True, False), bytes_=9 * n)
timeit("assign labels+dist", lambda: km.assign(0, True, False, True), bytes_=9 * n)
lab, _, d = km.assign(0, True, False, True)
timeit("top_keys m=150", lambda: km._top_keys(d, x, 150))
timeit("bincount", lambda: ops.bincount(lab, 256), bytes_=n)
timeit("iterate(1) (assign+finalize)", lambda: km.iterate(1))
