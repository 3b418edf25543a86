#!/usr/bin/env python3
"""The two passes over the vector of the k-means path on the bench vector (25 M weights pruned at 1 sigma, K = 257, converged centres),
each timed on its own by HIP events around the launch (in-library): the assignment pass k_assign<labels> (4 B read + 2 B index + 4 B
value per weight) and the streaming Lloyd pass k_assign<accumulate> (4 B read per weight)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat, kmeans, ops, pipeline, synth
L = nat.load()
n = 25_000_000
x = torch.from_numpy(synth.weights((n,), 4000)).cuda()
ops.prune_(x, 1.0, True)
res = pipeline.compress_layer(x.clone(), q=None, bits=8, mode="density", huffman=False, want_values=True)
centers = res.model.cluster_centers_.ravel()
def timed(tag, fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    nat.check(L.nnc_profile_tags(1 << tag)); nat.check(L.nnc_profile_begin(64))
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    ms, tg, c = (ctypes.c_float * 64)(), (ctypes.c_int32 * 64)(), ctypes.c_int64(0)
    nat.check(L.nnc_profile_end(ms, tg, 64, ctypes.byref(c))); nat.check(L.nnc_profile_tags(0xFFFFFFFF))
    return np.array([ms[i] for i in range(min(c.value, 64)) if tg[i] == tag]) * 1e3
km = kmeans.DeviceKMeans(x, centers, rank_boundaries=False)
lab = torch.empty(n, dtype=torch.int16, device="cuda"); val = torch.empty(n, dtype=torch.float32, device="cuda")
d = timed(2, lambda: nat.check(L.nnc_kmeans_assign(km.x.data_ptr(), km.ws.data_ptr(), ctypes.byref(km.p), 0, lab.data_ptr(), 2, val.data_ptr(), None, None, km.stream)))
print(f"k_assign<labels>: mean {d.mean():.2f} us, median {np.median(d):.2f}, min {d.min():.2f} -> {250e6 / d.mean() / 1e6 / 8:.3f} of 8 TB/s (10 B per weight), {100e6 / d.mean() / 1e6 / 8:.3f} by the 4 B read alone")
d = timed(0, lambda: nat.check(L.nnc_kmeans_accumulate(km.x_iter.data_ptr(), km.ws.data_ptr(), ctypes.byref(km.p), km.stream)))
print(f"k_assign<accumulate>: mean {d.mean():.2f} us, median {np.median(d):.2f}, min {d.min():.2f} -> {100e6 / d.mean() / 1e6 / 8:.3f} of 8 TB/s (4 B per weight)")
