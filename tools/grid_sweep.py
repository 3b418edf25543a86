"""Fit time against the cell-grid size for mid-sized layers with few centres (the table is copied into LDS by every workgroup)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, ops, synth
dev = torch.device("cuda:0")
for n, k in ((589_824, 16), (2_359_296, 16), (2_359_296, 64), (235_200, 32), (25_000_000, 256)):
    x = torch.from_numpy(synth.weights((n,), 4000)).to(dev)
    ops.prune_(x, 1.0, True)
    mm = ops.minmax(x)[0].cpu().numpy()
    init = np.linspace(mm[0], mm[1], k).astype(np.float32)
    for g in (14, 13, 12, 11, 10):
        ts = []
        for rep in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            km = kmeans.DeviceKMeans(x, init, grid_log2=g)
            m, _ = km.fit(want_values=False)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"n={n:>9} K={k:>3} grid 2^{g}: {np.median(ts[1:])*1e3:7.3f} ms  ({m.n_iter_} iterations)", flush=True)
