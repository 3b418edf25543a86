#!/usr/bin/env python3
"""Phase timeline of k_bounds (diagnostics build: NNC_DIAG=1): per wave {start, zones known, hint round done, searches done,
certain stretch added, undecided stretch done, helped, end} in 10 ns ticks, for one iteration of the bench fit."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat, kmeans, ops, pipeline, synth
L = nat.load()
n = 25_000_000
x = torch.from_numpy(synth.weights((n,), 4000)).cuda()
ops.prune_(x, 1.0, True)
st = kmeans.LayerStats(x)
xs = kmeans.sorted_copy(x, st)
cdfs = pipeline.weight_distribution_sorted(xs, st)
space = pipeline.initial_centroids(x, 8, "density", cdfs)
km = kmeans.DeviceKMeans(x, space, stats=st, x_sorted=xs, two_launch=True)
target = [int(a) for a in sys.argv[1:]] or [2, 30]
trace = torch.zeros(8192, dtype=torch.int64, pin_memory=True)
it = 0
while True:
    if it in target:
        cen_before = np.sort(np.unique(km.centers(centred=True)))
        nat.check(L.nnc_debug_set_trace(trace.data_ptr()))
    s = km.iterate_and_look(1)
    torch.cuda.synchronize()
    if it in target:
        nat.check(L.nnc_debug_set_trace(None))
        t = trace.numpy().reshape(-1, 16)[:260].copy()
        act = t[:, 0] > 0
        t0 = t[act, 0].min()
        rel = (t[act][:, :7] - t0) * 0.01
        und = t[act][:, 8]
        ncand = t[act][:, 9] + 1
        sure = t[act][:, 10]
        print(f"iteration {it}: {act.sum()} waves; start {np.median(rel[:,0]):.2f} us (max {rel[:,0].max():.2f}); median per phase (us since kernel start): "
              f"zones {np.median(rel[:,1]):.2f}, hint {np.median(rel[:,2]):.2f}, searches {np.median(rel[:,3]):.2f}, certain {np.median(rel[:,4]):.2f}, "
              f"undecided {np.median(rel[:,5]):.2f}, end {np.median(rel[:,6]):.2f}; last wave ends {rel[:,6].max():.2f}; undecided samples median {np.median(und):.0f} max {und.max()}")
        worst = np.argsort(rel[:, 6])[-3:]
        for w in worst:
            print("   slow wave", int(np.nonzero(act)[0][w]), [round(float(v), 2) for v in rel[w]], "undecided", int(und[w]), "candidates", int(ncand[w]), "certain", int(sure[w]))
        jm = int(np.argmax(und)); jj = int(np.nonzero(act)[0][jm])
        lo_, hi_ = max(0, jj - 2), min(len(cen_before), jj + 4)
        print("   longest stretch: wave", jj, "undecided", int(und[jm]), "candidates", int(ncand[jm]), "distinct centres", len(cen_before),
              "centres around it", [float(v) for v in cen_before[lo_:hi_]], "gaps", [float(v) for v in np.diff(cen_before[lo_:hi_])])
        print("   undecided total", int(und[:-4].sum()), "; waves with > 2 candidates:", int((ncand[:-4] > 2).sum()), "max candidates", int(ncand[:-4].max()))
        trace.zero_()
    if s.paused:
        km._relocate_and_resume(s)
    if s.done or it > max(target):
        break
    it = int(s.iter)
