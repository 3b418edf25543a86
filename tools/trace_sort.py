#!/usr/bin/env python3
"""Phases of a radix pass (k_os_pass), from timestamps thread 0 of every workgroup takes per tile (diagnostics build:
NNC_DIAG=1 python tools/trace_sort.py).  Bench vector: 8 M keys of 26 bits, three passes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat, ops, synth
L = nat.load()
n = 25_000_000
x = torch.from_numpy(synth.weights((n,), 4000)).cuda()
mask, stats, nz, mm, signs = ops.prune_stats_(x, 1.0, True)
thr = float(stats.cpu().numpy()[1]); mmh, sg = mm.cpu().numpy(), signs.cpu().numpy()
n_neg, n_zero = int(sg[0]), int(sg[1])
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty_like(x)
wsb = int(L.nnc_sort_pruned_bounded_workspace_bytes(n - n_zero)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
run = lambda: nat.check(L.nnc_sort_pruned_bounded_f32(x.data_ptr(), n, n_neg, n_zero, float(mmh[0]), float(mmh[1]), thr, out.data_ptr(), ws.data_ptr(), wsb, stream))
for _ in range(3): run()
tr = torch.zeros(4 * 8192 * 8, dtype=torch.int64, device="cuda")
nat.check(L.nnc_debug_os_trace(tr.data_ptr()))
run(); torch.cuda.synchronize()
nat.check(L.nnc_debug_os_trace(0))
t = tr.cpu().numpy().reshape(4, 8192, 8)
TILE = int(os.environ.get("OS_TILE", "8192"))
ntiles = (n - n_zero + TILE - 1) // TILE
names = ["claim->start", "load keys", "rank", "tile scan + LDS reorder", "look-back", "barrier", "write out", "barrier"]
for p in range(3):
    a = t[p, :ntiles].astype(np.float64) * 0.01   # us
    t0 = a[:, 0].min()
    d = np.diff(a, axis=1)
    print(f"pass {p}: {ntiles} tiles; first tile starts 0.0, last tile ends {a[:, 7].max() - t0:.1f} us; per tile total median {np.median(a[:, 7] - a[:, 0]):.1f} us")
    print("   phase medians (us): " + ", ".join(f"{nm} {np.median(d[:, i]):.2f}" for i, nm in enumerate(names[1:])))
    print("   phase p90 (us):     " + ", ".join(f"{nm} {np.percentile(d[:, i], 90):.2f}" for i, nm in enumerate(names[1:])))
    order = np.argsort(a[:, 0])
    print("   tile start times (us) of tiles 0, 1, 255, 511, 512, 900:", [round(float(a[i, 0] - t0), 1) for i in (0, 1, 255, 511, 512, min(900, ntiles - 1))])
    print("   look-back by tile index (us): ", [round(float(d[i, 3]), 1) for i in (1, 2, 8, 64, 128, 256, 400, 511, 600, min(900, ntiles - 1))])
# where the look-back time goes: a tile can finish its look-back once every tile in front has published its counts (stamp 2 of
# that tile, roughly); what lies between that moment (or the tile's own arrival at the look-back, if later) and the end of the
# look-back is the walk + the time the status words take to become visible
for p in range(3):
    a = t[p, :ntiles].astype(np.float64) * 0.01
    pub = np.maximum.accumulate(a[:, 2])            # all tiles <= k have published by then
    ready = np.concatenate([[a[0, 3]], np.maximum(pub[:-1], a[1:, 3])])
    lag = a[:, 4] - ready
    waited = np.maximum(pub[:-1] - a[1:, 3], 0)
    print(f"pass {p}: waiting for the tiles in front to publish: median {np.median(waited):.2f} us, p90 {np.percentile(waited, 90):.2f}; "
          f"from then to the end of the look-back: median {np.median(lag):.2f} us, p90 {np.percentile(lag, 90):.2f}")
