"""cProfile of the host side of bench steps (where the Python time goes)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import pipeline, synth
dev = torch.device("cuda:0")
w0 = torch.from_numpy(synth.weights((25_000_000,), 4000)).to(dev)
def step():
    return pipeline.compress_layer(w0.clone(), q=1.0, bits=8, mode="density", huffman=True, want_values=True)
step(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
