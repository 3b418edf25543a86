#!/usr/bin/env python3
"""Time of the value-sorted copy, by HIP events: the three entry points of csrc/nnc_sort.hip on the bench vector (25 M weights, pruned
at 1 sigma: 8 M keys of 26 bits / of 32 bits) and on the unpruned vector (25 M keys of 32 bits)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat, ops, synth
L = nat.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
w = synth.weights((n,), 4000)
raw = torch.from_numpy(w).cuda()
x = raw.clone()
mask, stats, nz, mm, signs = ops.prune_stats_(x, 1.0, True)
thr = float(stats.cpu().numpy()[1]); mmh, sg = mm.cpu().numpy(), signs.cpu().numpy()
n_neg, n_zero = int(sg[0]), int(sg[1])
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty_like(x)
def timed(name, fn, check_against):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ok = torch.equal(out, check_against)
    print(f"{name}: median {np.median(ts):.1f} us, min {min(ts):.1f} us, correct {ok}", flush=True)
want_p = torch.sort(x).values
wsb = int(L.nnc_sort_pruned_bounded_workspace_bytes(n - n_zero)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
timed(f"bounded pruned sort, {n} weights, {n - n_zero} keys of {L.nnc_sort_pruned_bounded_bits(float(mmh[0]), float(mmh[1]), thr, n_neg, n - n_neg - n_zero)} bits",
      lambda: nat.check(L.nnc_sort_pruned_bounded_f32(x.data_ptr(), n, n_neg, n_zero, float(mmh[0]), float(mmh[1]), thr, out.data_ptr(), ws.data_ptr(), wsb, stream)), want_p)
wsb2 = int(L.nnc_sort_pruned_workspace_bytes(n, n_neg, n_zero)); ws2 = torch.empty(wsb2, dtype=torch.uint8, device="cuda")
timed(f"pruned sort without bounds, {n - n_zero} keys of 32 bits",
      lambda: nat.check(L.nnc_sort_pruned_f32(x.data_ptr(), n, n_neg, n_zero, out.data_ptr(), ws2.data_ptr(), wsb2, stream)), want_p)
want_u = torch.sort(raw).values
wsb3 = int(L.nnc_sort_workspace_bytes(n)); ws3 = torch.empty(wsb3, dtype=torch.uint8, device="cuda")
timed(f"plain sort of the unpruned vector, {n} keys of 32 bits",
      lambda: nat.check(L.nnc_sort_f32(raw.data_ptr(), n, out.data_ptr(), ws3.data_ptr(), wsb3, stream)), want_u)
