"""cProfile of the host side of a tiny layer (n = 768, K = 16, linear init): where the fixed cost of a fit goes."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import pipeline, synth
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
w0 = torch.from_numpy(synth.weights((n,), 4000)).to(dev)
def step():
    return pipeline.compress_layer(w0.clone(), q=1.0, bits=4, mode="linear", huffman=True, want_values=True)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): r = step()
torch.cuda.synchronize()
print(f"n={n}: {(time.perf_counter()-t0)/50*1e3:.3f} ms per layer, {r.model.n_iter_} iterations, {r.model.n_relocations_} relocations")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(30)
