import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, ops, pipeline, synth
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.weights((25_000_000,), 4000)).to(dev)
ops.prune_(x, 1.0, True)
cdfs = pipeline.weight_distribution(x, True)
space = pipeline.initial_centroids(x, 8, "density", cdfs)
km = kmeans.DeviceKMeans(x, space)
for it in range(1, 36):
    km.iterate(1); st = km.status()
    if st.paused:
        km._relocate_and_resume(st); st = km.status()
    if it >= 28:
        c0 = km.centers(0, True); c1 = km.centers(1, True)
        o0 = np.lexsort((np.arange(km.k), c0)); o1 = np.lexsort((np.arange(km.k), c1))
        print(it, "same order:", np.array_equal(o0, o1), "distinct", len(np.unique(c0)), "ndiff", int((o0 != o1).sum()))
