#!/usr/bin/env python3
"""Average duration of k_assign<accumulate> over one fit of the bench workload (in-library HIP events).
   ABL=<n> selects an experimental kernel variant (nnc_debug_set_ablation); PRUNE=0 for the dense vector; GRID=<log2> the cell grid."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat, kmeans, ops, pipeline, synth
L = nat.load()
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.weights((25_000_000,), 4000)).to(dev)
pruned = os.environ.get("PRUNE", "1") == "1"
if pruned:
    ops.prune_(x, 1.0, True)
cdfs = pipeline.weight_distribution(x, pruned)
space = pipeline.initial_centroids(x, 8, "density", cdfs)
for abl in [int(a) for a in os.environ.get("ABL", "0").split(",")]:
    nat.check(L.nnc_debug_set_ablation(abl))
    res = []
    for rep in range(3):
        km = kmeans.DeviceKMeans(x, space, grid_log2=int(os.environ.get("GRID", "0")))
        nat.check(L.nnc_profile_begin(400))
        model, _ = km.fit(False)
        torch.cuda.synchronize()
        buf = (ctypes.c_float * 400)(); cnt = ctypes.c_int64(0)
        nat.check(L.nnc_profile_end(buf, 400, ctypes.byref(cnt)))
        d = np.array(buf[: cnt.value]) * 1e3
        live = d[d > 0.3 * np.median(d)]
        res.append((live.mean(), np.median(live), live.min(), len(live), model.n_iter_))
    print(f"ABL {abl}: " + "; ".join(f"avg {a:.2f} med {m:.2f} min {mn:.2f} us ({n} launches, n_iter {it})" for a, m, mn, n, it in res))
nat.check(L.nnc_debug_set_ablation(0))
