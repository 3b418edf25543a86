#!/usr/bin/env python3
"""Repeated prune -> 4-bit linear-init k-means -> Huffman on one short tensor (BASELINE configs[4]'s 768-weight layers):
wall time per fit; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t = torch.from_numpy(synth.weights((n,), 5011)).cuda()
for _ in range(5):
    r = pipeline.compress_layer(t.clone(), q=1.0, bits=4, mode="linear")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    r = pipeline.compress_layer(t.clone(), q=1.0, bits=4, mode="linear")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"n={n}: {dt / reps * 1e3:.3f} ms per fit, {r.model.n_iter_} iterations, {r.model.n_relocations_} relocations")
