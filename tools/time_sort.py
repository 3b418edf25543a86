#!/usr/bin/env python3
"""Time of the sorted copy of the bench vector's 8 M surviving weights (nnc_sort_pruned_bounded_f32), by HIP events."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, ops, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
x = torch.from_numpy(synth.weights((n,), 4000)).cuda()
ops.prune_(x, 1.0, True)
st = kmeans.LayerStats(x)
for _ in range(3):
    xs = kmeans.sorted_copy(x, st)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); xs = kmeans.sorted_copy(x, st); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
assert bool((xs[1:] >= xs[:-1]).all())
print("sorted copy of %d weights (%d non-zero): median %.1f us, min %.1f us" % (n, int((x != 0).sum()), float(np.median(ts)), min(ts)))
