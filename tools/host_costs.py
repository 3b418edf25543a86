"""Host-side costs of the fit loop: status read latency, enqueue time of one iteration / one relocation."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans, ops, pipeline, synth
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.weights((25_000_000,), 4000)).to(dev)
ops.prune_(x, 1.0, True)
cdfs = pipeline.weight_distribution(x, True)
space = pipeline.initial_centroids(x, 8, "density", cdfs)
km = kmeans.DeviceKMeans(x, space)
torch.cuda.synchronize()
def t(f, n=200):
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - a) / n * 1e6
print(f"status() on an idle stream: {t(km.status):.1f} us")
km.iterate(1); st = km.status(); print("paused", st.paused)
# enqueue cost of no-op iterations (state is paused: kernels return at once)
torch.cuda.synchronize(); a = time.perf_counter()
for _ in range(100): km.iterate(1)
b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
print(f"iterate(1) enqueue {1e4*(b-a):.1f} us each; drain {1e6*(c-b):.0f} us for 100 no-op iterations")
a = time.perf_counter(); km._relocate_windowed(int(st.n_empty)); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
print(f"_relocate_windowed enqueue {1e6*(b-a):.0f} us, then wait {1e6*(c-b):.0f} us")
def it_sync():
    km.iterate(1); km.status()
print(f"iterate(1)+status: {t(it_sync, 20):.1f} us")
