"""Where the iterations of the bench fit run (loop vs wide pair), and the time per fit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neural_network_compression_amd import kmeans as km, pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
w = torch.from_numpy(synth.weights((n,), 4000)).cuda()
x = w.clone()
res = pipeline.compress_layer(x, q=1.0, bits=8, mode="density", huffman=True, want_values=True, native=False)
print("step-by-step: n_iter", res.model.n_iter_, "reloc", res.model.n_relocations_)
# the same fit through DeviceKMeans, to read the counters
x = w.clone()
pipeline.prune_sharded_(x, 1.0, True, None)
from neural_network_compression_amd.common import utility as U
cdfs = U.get_weight_distribution(x, skip_zeros=True)
space = np.asarray(U._init_space(x, x.numel(), 8, "density", cdfs), dtype=np.float32)
for tl in (False, False, True, True):
    d = km.DeviceKMeans(x, space, two_launch=tl, loop=not tl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m, _ = d.fit()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("two_launch", tl, "n_iter", m.n_iter_, "reloc", m.n_relocations_, "windowed", m.n_reloc_windowed_, "fit ms %.3f" % (dt * 1e3), d.loop_stats())
