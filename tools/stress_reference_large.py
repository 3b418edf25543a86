#!/usr/bin/env python3
"""Randomised stress of the opt-in reference-arithmetic fit beyond 4096 weights (kmeans.fit_reference_large: float32 sums in sample
order on the device, numpy.argpartition relocation) against the oracle's mode A (scikit-learn on one thread, restated), bit for bit."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, synth
from oracle import oracle as orc

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
t0 = time.time()
for case in range(ncases):
    n = int(rng.choice([4_097, 5_000, 20_000, 66_000, 131_072, 235_200, 300_000]))
    x = synth.weights((n,), 7000 + case, scale=float(rng.choice([0.05, 0.5, 3e-4])))
    kind = rng.randint(0, 4)
    if kind in (0, 1):
        x[np.abs(x) < np.float32(rng.uniform(0.2, 1.5)) * x.std()] = 0
    if kind == 2:
        x = (np.round(x / x.std() * rng.randint(3, 40)) * x.std() / 17).astype(np.float32)
    k = int(rng.choice([2, 4, 16, 17, 32, 33, 64, 257]))
    style = rng.randint(0, 4)
    lo, hi = float(x.min()), float(x.max())
    if style == 0:
        init = np.linspace(lo, hi, k)
    elif style == 1:
        init = x[rng.randint(0, n, size=k)]                      # forgy: duplicates -> relocations
    elif style == 2:
        init = np.linspace(lo * 1.7, hi * 1.7, k)                # centres outside the data -> empty clusters
    else:
        init = np.concatenate([np.full(k // 2, np.median(x)), rng.uniform(lo, hi, k - k // 2)])
    init = np.asarray(init, dtype=np.float32)
    oa = orc.kmeans_lloyd(x, init, accum="A")
    model, vals = kmeans.fit_reference_large(torch.from_numpy(x).cuda(), init)
    ok = (model.n_iter_ == oa.n_iter_ and np.array_equal(model.cluster_centers_.ravel().view(np.uint32), oa.cluster_centers_.ravel().view(np.uint32))
          and np.array_equal(model.labels_, oa.labels_) and np.array_equal(vals.cpu().numpy(), oa.cluster_centers_.ravel()[oa.labels_]))
    print(f"case {case}: n={n} k={k} kind={kind} init={style} n_iter={model.n_iter_}/{oa.n_iter_} reloc={model.n_relocations_} "
          f"ties={model.reloc_tie_} stop={model.stop_reason_} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += (not ok)
print(f"seed {sys.argv[1] if len(sys.argv) > 1 else 1}: {ncases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
