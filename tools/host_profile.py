#!/usr/bin/env python3
"""Where the host time of a bench step goes (cProfile over 20 steps of pipeline.compress_layer on the bench vector)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = sys.argv[3] if len(sys.argv) > 3 else "density"
w0 = torch.from_numpy(synth.weights((n,), 4000)).cuda()
def step():
    return pipeline.compress_layer(w0.clone(), q=1.0, bits=bits, mode=mode, huffman=True, want_values=True)
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
