#!/usr/bin/env python3
"""Time of the one-launch reference-arithmetic fit (nnc_kmeans_fit_reference_f32) per size: kernel time per Lloyd iteration."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, ops, synth

for n, k, prune in [(768, 16, True), (2304, 16, True), (3072, 16, True), (4096, 16, True), (4096, 16, False), (4096, 32, False), (4096, 128, False), (1000, 65, True)]:
    w = synth.weights((n,), 5000 + n)
    t = torch.from_numpy(w).cuda()
    if prune:
        ops.prune_(t, 1.0, True)
    mm = t.cpu().numpy()
    init = np.linspace(mm.min(), mm.max(), k).astype(np.float32)
    m, _ = kmeans.fit_reference(t, init)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        m, _ = kmeans.fit_reference(t, init)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"n={n:5d} k={k:3d} pruned={int(prune)}: {m.n_iter_:3d} iterations, {m.n_relocations_} relocations, {us:8.1f} us per fit (host included), {us / m.n_iter_:6.1f} us per iteration")
    pt = getattr(m, "phase_times_", None)
    if os.environ.get("NNC_DIAG", "0") not in ("", "0") and pt is not None:
        names = ["setup", "E-step", "prefix", "place+sums", "relocation", "average", "epilogue"]
        print("      phases (us per fit): " + ", ".join(f"{nm} {pt[i] / 100.0:.1f}" for i, nm in enumerate(names)))
