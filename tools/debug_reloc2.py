import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import _native as nat, kmeans, ops, pipeline, synth
L = nat.load()
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.weights((25_000_000,), 4000)).to(dev)
ops.prune_(x, 1.0, True)
cdfs = pipeline.weight_distribution(x, True)
space = pipeline.initial_centroids(x, 8, "density", cdfs)
km = kmeans.DeviceKMeans(x, space)
km.iterate(1); st = km.status()
print("paused", st.paused, "n_empty", st.n_empty)
lab = km._assign_on(km.x_iter, which=0, labels=True)[0].cpu().numpy().astype(np.int64) & 0xFFFF
xs = km.x_iter.cpu().numpy()
cen = km.centers(which=0, centred=True)
counts = np.bincount(lab, minlength=km.k)
part = km.partials.cpu().numpy()
print("counts equal partials:", np.array_equal(counts, part[km.k:]))
# distinct sorted centres and their lowest original index
order = np.lexsort((np.arange(km.k), cen))
cs = cen[order]
first = np.r_[True, cs[1:] != cs[:-1]]
orig = order[first]; cu = cs[first]
cnt_sorted = counts[orig]
print("distinct", len(cu), "sum", cnt_sorted.sum(), "n", xs.size)
B = np.r_[0, np.cumsum(cnt_sorted)]
# actual label (as sorted-distinct index) along the sorted vector: is it monotone?
inv = np.full(km.k, -1); inv[orig] = np.arange(len(orig))
ls = inv[lab]
print("labels all map to distinct reps:", (ls >= 0).all(), " monotone:", bool(np.all(np.diff(ls) >= 0)))
bad = np.nonzero(np.diff(ls) < 0)[0]
print("non-monotone places:", bad[:10], [(xs[i], xs[i+1], ls[i], ls[i+1]) for i in bad[:5]])
for j in range(1, len(cu)):
    b = B[j]
    if cnt_sorted[j-1] and cnt_sorted[j] and not (ls[b-1] == j-1 and ls[b] == j):
        print("boundary mismatch at", j, b, ls[b-2:b+2]); break
