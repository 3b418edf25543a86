"""Fit time of the one-workgroup loop against the launch-per-iteration form over (n, K): where does each win?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neural_network_compression_amd import kmeans as km, pipeline, synth
from neural_network_compression_amd.common import utility as U
for n in (100_000, 600_000, 2_400_000, 10_000_000, 25_000_000):
    x = torch.from_numpy(synth.weights((n,), 4000 + n % 97)).cuda()
    pipeline.prune_sharded_(x, 1.0, True, None)
    for bits, mode in ((4, "linear"), (5, "linear"), (6, "density"), (7, "density"), (8, "density")):
        cdfs = U.get_weight_distribution(x, skip_zeros=True) if mode == "density" else None
        space = np.asarray(U._init_space(x, x.numel(), bits, mode, cdfs), dtype=np.float32)
        row = []
        for two in (False, True):
            best = 1e9
            for rep in range(3):
                d = km.DeviceKMeans(x, space, two_launch=two)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                m, _ = d.fit()
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            row.append((best * 1e3, m.n_iter_, m.n_relocations_))
        print(f"n={n:>9} K={space.size:4d}: loop {row[0][0]:7.3f} ms  two-launch {row[1][0]:7.3f} ms  ({row[0][1]} iterations, {row[0][2]} relocations)  ratio {row[0][0] / row[1][0]:.2f}", flush=True)
