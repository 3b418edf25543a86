#!/usr/bin/env python3
"""Diagnostics build (NNC_DIAG=1): the empty-cluster events of the bench fit as the finalize step saw them, and the phases of the
ones it settled itself as mass events (kl_relocate_mass)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat
from neural_network_compression_amd import kmeans, ops, pipeline, synth
dev = torch.device("cuda:0")
L = nat.load()
n = 25_000_000
x = torch.from_numpy(synth.weights((n,), 4000)).to(dev)
ops.prune_(x, 1.0, True)
cdfs = pipeline.weight_distribution(x, True)
space = pipeline.initial_centroids(x, 8, "density", cdfs)
tr = torch.zeros(8192, dtype=torch.int64, device=dev)
km = kmeans.DeviceKMeans(x, space, two_launch=True)
nat.check(L.nnc_debug_set_trace(tr.data_ptr()))
m, _ = km.fit()
torch.cuda.synchronize()
nat.check(L.nnc_debug_set_trace(0))
t = tr.cpu().numpy()[6 * 1024:]
print("fit:", m.n_iter_, "iterations,", m.n_relocations_, "events")
for it in range(64):
    ne, nch, slow, ku = (int(v) for v in t[300 + 4 * it: 304 + 4 * it])
    if ne:
        print(f"  iteration {it}: {ne} empty clusters, {ku} distinct centres, {nch} chunks of undecided samples, long stretch {slow}")
names = ["ends", "first selection", "second turn", "undecided", "final rank + proof + empties", "old clusters", "edits"]
for o in range(16):
    s = t[100 + 10 * o: 110 + 10 * o]
    if s[0] == 0:
        continue
    d = np.diff(s[:8].astype(np.float64)) * 0.01
    N, ms = int(s[8]) >> 32, int(s[8]) & 0xFFFFFFFF
    done = s[7] > 0
    print(f"  mass event #{o}: m = {int(s[9])}, {N} candidate keys in the first turn, {ms} ends into the second; " + (", ".join(f"{nm} {v:.1f}" for nm, v in zip(names, d)) if done else "not settled (phases reached: " + ", ".join(f"{nm} {v:.1f}" for nm, v in zip(names, d) if v > 0) + ")"))
# the LAST event the finalize step settled in place (the stamps of later launches do not touch these slots): phases since the launch's start
if t[22] > 0:
    z = float(t[20])
    r = (t[30:38].astype(np.float64) - z) * 0.01
    print(f"last event settled in place: selection starts {(t[21] - z) * 0.01:.1f}, certain ends {r[1]:.1f}, undecided samples {r[2]:.1f}, barrier {r[3]:.1f}, keys in registers {r[4]:.1f}, "
          f"rounds {r[5]:.1f}, proof + empties {r[6]:.1f}, old clusters + edits {r[7]:.1f}, selection done {(t[22] - z) * 0.01:.1f}; shift / tolerance {(t[23] - z) * 0.01:.1f}, centres sorted {(t[24] - z) * 0.01:.1f}, "
          f"zones {(t[25] - z) * 0.01:.1f}, end {(t[26] - z) * 0.01:.1f} us")
