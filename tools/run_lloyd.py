#!/usr/bin/env python3
"""Profiling driver: N Lloyd iterations of the streaming kernel on a resident 25 M vector.
usage: python tools/run_lloyd.py [--n 25000000] [--k 256] [--iters 20] [--pruned] [--grid-log2 G] [--rep-log2 R]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=25_000_000)
ap.add_argument("--k", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--pruned", action="store_true")
ap.add_argument("--grid-log2", type=int, default=0)
ap.add_argument("--rep-log2", type=int, default=-1)
ap.add_argument("--torch-randn", action="store_true")
ap.add_argument("--ablation", type=int, default=0)
ap.add_argument("--no-sort", action="store_true")
ap.add_argument("--accum-only", action="store_true")
a = ap.parse_args()

dev = torch.device("cuda:0")
if a.torch_randn:
    x = torch.randn(a.n, device=dev) * 0.05
else:
    x = torch.from_numpy(synth.weights((a.n,), 4000)).to(dev)
if a.pruned:
    ops.prune_(x, 1.0, True)
# quantile init: no duplicates, no empty clusters
xs = x[:: max(1, a.n // 1_000_000)].float()
nz = xs[xs != 0] if a.pruned else xs
qs = torch.quantile(nz[:1_000_000], torch.linspace(0.0005, 0.9995, a.k, device=dev)).cpu().numpy()
km = kmeans.DeviceKMeans(x, np.unique(qs).astype(np.float32) if False else qs.astype(np.float32), max_iter=10_000, tol=0.0,
                         grid_log2=a.grid_log2, replicas_log2=a.rep_log2, sort=not a.no_sort)
km.iterate(3)
st = km.status()
torch.cuda.synchronize()
import ctypes
from neural_network_compression_amd import _native as nat
L = nat.load()
nat.check(L.nnc_debug_set_ablation(a.ablation))
nat.check(L.nnc_profile_begin(a.iters + 8))
t0 = time.perf_counter()
if a.accum_only:
    for _ in range(a.iters):
        nat.check(L.nnc_kmeans_accumulate(km.x_iter.data_ptr(), km.ws.data_ptr(), ctypes.byref(km.p), km.stream))
else:
    km.iterate(a.iters)
st = km.status()
dt = time.perf_counter() - t0
buf = (ctypes.c_float * (a.iters + 8))()
cnt = ctypes.c_int64(0)
nat.check(L.nnc_profile_end(buf, a.iters + 8, ctypes.byref(cnt)))
d = np.array(buf[: cnt.value])
print(f"  k_assign<accumulate>: {cnt.value} launches, median {np.median(d)*1e3:.1f} us, min {d.min()*1e3:.1f} us -> "
      f"{4 * a.n / (np.median(d) * 1e-3) / 1e9:.0f} GB/s = {4 * a.n / (np.median(d) * 1e-3) / 8e12 * 100:.1f} % of 8 TB/s")
print(f"n={a.n} k={a.k} pruned={a.pruned} iters={st.iter} done={st.done} paused={st.paused} "
      f"{dt / a.iters * 1e6:.1f} us/iter (kernel+finalize, host-inclusive) -> {4 * a.n / (dt / a.iters) / 1e9:.0f} GB/s algorithmic")
