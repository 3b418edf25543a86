#!/usr/bin/env python3
"""Diagnostics build (NNC_DIAG=1): the relocation chain enqueued behind an iteration, one launch at a time with a
synchronisation in between, on the bench fit."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat, kmeans, ops, pipeline, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
x = torch.from_numpy(synth.weights((n,), 4000)).cuda()
ops.prune_(x, 1.0, True)
ls = kmeans.LayerStats(x, n, None)
xs = kmeans.sorted_copy(x, ls)
cdfs = pipeline.weight_distribution_sorted(xs, ls, None)
space = pipeline.initial_centroids(x, 8, "density", cdfs, None, n)
km = kmeans.DeviceKMeans(x, space, stats=ls, x_sorted=xs, n_total=n, n_min=n)
L = km.L
need = int(L.nnc_kmeans_reloc_scratch_bytes(km.k, 256))
scratch = torch.empty(need, dtype=torch.uint8, device=x.device)
for it in range(60):
    km.iterate(1)
    torch.cuda.synchronize()
    st = km.status()
    print(f"iteration {it}: iter {st.iter} done {st.done} paused {st.paused} n_empty {st.n_empty} relocated {st.n_relocated}", flush=True)
    if st.done:
        break
    for stage, name in [(1, "windows"), (2, "cells"), (4, "dist"), (8, "select"), (16, "finalize")]:
        nat.check(L.nnc_debug_spec_stage(km.x_iter.data_ptr(), km.ws.data_ptr(), ctypes.byref(km.p), scratch.data_ptr(), stage, km.stream))
        torch.cuda.synchronize()
        print(f"   {name} ok", flush=True)
    st = km.status()
    print(f"   -> iter {st.iter} paused {st.paused} relocated {st.n_relocated}", flush=True)
    if st.paused:
        print("   still paused: host path")
        km._relocate_and_resume(st)
