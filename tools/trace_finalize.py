#!/usr/bin/env python3
"""Phase stamps of k_finalize (thread 0) and of the k_bounds waves (lane 0 of each) for a small-K and the headline fit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat
from neural_network_compression_amd import kmeans, ops, pipeline, synth

dev = torch.device("cuda:0")
L = nat.load()
order = [(0, "start"), (1, "shards->partials"), (2, "empties+average"), (3, "shift+tol+state"), (12, "perm filled"), (8, "still-sorted"), (9, "rank sort"),
         (10, "distinct"), (4, "tables"), (13, "pair zones"), (14, "barrier"), (15, "own zone + stores"), (11, "zones raw"), (5, "zone scans"), (7, "end")]
for n, bits, mode in ((2_359_296, 4, "linear"), (25_000_000, 8, "density")):
    x = torch.from_numpy(synth.weights((n,), 4000)).to(dev)
    ops.prune_(x, 1.0, True)
    cdfs = pipeline.weight_distribution(x, True) if mode == "density" else None
    space = pipeline.initial_centroids(x, bits, mode, cdfs)
    km = kmeans.DeviceKMeans(x, space, two_launch=True)
    tr = torch.zeros(8192, dtype=torch.int64, device=dev)
    acc = []
    late = []
    bnd = []   # per plain iteration: the k_bounds waves' stamps relative to the first wave's start, and the gap to the finalize step
    for it in range(1, 60):
        tr.zero_()
        if it >= 1:
            nat.check(L.nnc_debug_set_trace(tr.data_ptr()))
        km.iterate(1)
        st = km.status()
        torch.cuda.synchronize()
        nat.check(L.nnc_debug_set_trace(0))
        if 1 <= it <= 14 and n > 20_000_000:   # the passes behind the placements of the first iterations: what their slowest waves did
            full = tr.cpu().numpy()
            kball = full[: 16 * 300].reshape(300, 16)
            live = (kball[:, 0] > 0) & (kball[:, 6] > 0)
            if live.any():
                kb = kball[live].astype(np.float64)
                t0 = kb[:, 0].min()
                end = (kb[:, 6] - t0) * 0.01
                order_ = np.argsort(-end)[:3]
                desc = []
                for o in order_:
                    r = (kb[o, :8] - t0) * 0.01
                    desc.append(f"[start {r[0]:.1f} zones {r[1]:.1f} hint {r[2]:.1f} search {r[3]:.1f} prefix {r[4]:.1f} own {r[5]:.1f} waited {r[7]:.1f} ({int(kb[o, 12])} looks) help {r[6]:.1f}; undecided {int(kb[o, 8])}, candidates {int(kb[o, 9]) + 1}]")
                und = kb[:, 8]
                w7 = (kb[:, 7] - t0) * 0.01
                print(f"   pass {it}: {int(live.sum())} waves, pass ends {end.max():.1f} us after its first wave starts (median wave {np.median(end):.1f}; own work done: median {np.median((kb[:, 5] - t0) * 0.01):.1f}, last {((kb[:, 5] - t0) * 0.01).max():.1f}; wait over: median {np.median(w7):.1f}, last {w7.max():.1f}; looks: median {int(np.median(kb[:, 12]))}, most {int(kb[:, 12].max())}); undecided samples {int(und.sum())} (largest {int(und.max())}, "
                      f"waves with more than 256: {int((und > 256).sum())}, more than 2048: {int((und > 2048).sum())}), most candidates {int(kb[:, 9].max()) + 1}; slowest waves: " + " ".join(desc))
        if it >= 6 and not st.paused:
            full = tr.cpu().numpy()
            t = full[6 * 1024: 6 * 1024 + 48]
            if t[7] > t[0]:
                (late if it >= 22 else acc).append((t - t[0]) * 0.01)
                kball = full[: 16 * 300].reshape(300, 16)
                live = (kball[:, 0] > 0) & (kball[:, 6] > 0)
                kb = kball[live][:, :7]
                if it >= 22 and len(kb):
                    t0 = kb[:, 0].min()
                    rel = (kb - t0) * 0.01
                    bnd.append(np.concatenate([np.median(rel, axis=0), rel.max(axis=0), [(t[0] - kb[:, 6].max()) * 0.01, (t[0] - t0) * 0.01, len(kb)]]))
        if st.done:
            break
        if st.paused:
            km._relocate_and_resume(st)
    f = np.median(np.array(acc), axis=0)
    print(f"n={n} K={2**bits}: k_finalize stamps, iterations 6-21, median of {len(acc)} (us since start): " + ", ".join(f"{nm} {f[i]:.2f}" for i, nm in order))
    if bnd:
        b = np.median(np.array(bnd), axis=0)
        names = ["start", "zones+phi", "hint round", "searches", "prefix+own", "wave done", "help done"]
        print(f"n={n}: k_bounds waves, iterations 22-, median over {len(bnd)} passes of (median wave | slowest wave) us since the first wave's start: "
              + ", ".join(f"{nm} {b[i]:.2f}|{b[7 + i]:.2f}" for i, nm in enumerate(names))
              + f"; last wave's end -> finalize start {b[14]:.2f}; first wave's start -> finalize start {b[15]:.2f}; waves {b[16]:.0f}")
    if late:
        f = np.median(np.array(late), axis=0)
        print(f"   inside shards->partials (us since start): header read + queue zeroed {f[44]:.2f}, sums zeroed + barrier {f[45]:.2f}, shards added into LDS + barrier {f[46]:.2f}, partials stored + barrier {f[47]:.2f}")
        print(f"n={n} K={2**bits}: k_finalize stamps, iterations 22-, median of {len(late)} (us since start): " + ", ".join(f"{nm} {f[i]:.2f}" for i, nm in order))
