#!/usr/bin/env python3
"""Randomised parity stress of the one-launch reference-arithmetic fit (kmeans.fit_reference) against the oracle's mode A
(scikit-learn's summation order, pinned on the reference's outputs) with the device's relocation order: bit for bit."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, synth
from oracle import oracle as orc

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = nrel = nmulti = nties = 0
t0 = time.time()
for case in range(ncases):
    n = int(rng.choice([5, 17, 63, 64, 65, 100, 255, 256, 257, 600, 768, 1000, 1023, 1024, 1025, 2304, 3000, 3072, 4000, 4095, 4096]))
    x = synth.weights((n,), 19000 + case, scale=float(rng.choice([0.05, 0.5, 3e-4, 20.0])))
    kind = rng.randint(0, 6)
    if kind in (0, 1):
        x[np.abs(x) < np.float32(rng.uniform(0.2, 1.5)) * x.std()] = 0
    if kind == 2:
        x = (np.round(x / max(x.std(), 1e-30) * rng.randint(2, 12)) * x.std() / 7).astype(np.float32)   # few distinct values: ties
    if kind == 3:
        x = x + np.float32(rng.uniform(-3, 3))       # mean far from zero
    k = int(min(n, rng.choice([1, 2, 3, 4, 8, 16, 17, 32, 33, 64, 65, 100, 128])))
    style = rng.randint(0, 5)
    lo, hi = float(x.min()), float(x.max())
    if style == 0:
        init = np.linspace(lo, hi, k)
    elif style == 1:
        init = np.repeat(np.quantile(x.astype(np.float64), np.linspace(0.02, 0.98, max(2, k // 4))), 4)[:k]
    elif style == 2:
        init = x[rng.randint(0, n, size=k)]
    elif style == 3:
        init = np.linspace(lo * 1.7 - 0.1, hi * 1.7 + 0.1, k)
    else:
        init = np.concatenate([np.full(k // 2, np.median(x)), rng.uniform(lo, hi, k - k // 2)])
    init = np.asarray(init, dtype=np.float32)
    if init.size < k:
        init = np.concatenate([init, np.full(k - init.size, init[-1], dtype=np.float32)])
    oa = orc.kmeans_lloyd(x, init, accum="A", reloc="descending")
    model, vals = kmeans.fit_reference(torch.from_numpy(x).cuda(), init)
    ok = (model.n_iter_ == oa.n_iter_ and np.array_equal(model.cluster_centers_.ravel(), oa.cluster_centers_.ravel())
          and np.array_equal(model.labels_, oa.labels_) and np.array_equal(vals.cpu().numpy(), oa.cluster_centers_.ravel()[oa.labels_])
          and np.array_equal(model.counts_host_, np.bincount(oa.labels_, minlength=k))
          and model.n_relocations_ == oa.reloc_info_.get("reloc_events", 0) and model.n_reloc_multi_ == oa.reloc_info_.get("reloc_multi", 0))
    nrel += model.n_relocations_; nmulti += model.n_reloc_multi_; nties += model.reloc_tie_
    if not ok or case % 25 == 0:
        print(f"case {case}: n={n} k={k} kind={kind} init={style} n_iter={model.n_iter_}/{oa.n_iter_} reloc={model.n_relocations_} stop={model.stop_reason_} "
              f"{'ok' if ok else 'MISMATCH'}", flush=True)
    bad += (not ok)
print(f"{ncases} cases, {bad} mismatches, {nrel} relocation events ({nmulti} with several empty clusters, {nties} with a tie at the cut), {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
