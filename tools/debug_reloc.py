"""Why does the windowed relocation proof fail?  (bench workload; prints the reason bits per event)"""
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
for it in range(60):
    km.iterate(1); st = km.status()
    if st.done: break
    if st.paused:
        ne = int(st.n_empty)
        ok = km._relocate_windowed(ne)
        torch.cuda.synchronize()
        r = ctypes.c_int32(0)
        nat.check(L.nnc_debug_reloc_fail(km.ws.data_ptr(), ctypes.byref(r)))
        st2 = km.status()
        print(f"iter {st.iter} n_empty {ne} same_counts {st.same_counts}: fail bits {r.value} paused {st2.paused}")
        if st2.paused:
            km._relocate_and_resume(st2)
