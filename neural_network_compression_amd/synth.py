"""Deterministic synthetic weight tensors (integer-only generator).

The parity tests, the golden-vector script and ``bench.py`` must see bit-identical
inputs in this container and on the GPU box, whatever the libm / numpy build.  A
Box-Muller generator depends on ``log``/``cos`` last-bit behaviour, so this one uses
integer arithmetic only:

* counter-based splitmix64 hash  ->  twelve 32-bit uniforms per element,
* their sum (Irwin-Hall, n = 12) has mean 6*(2^32-1) and standard deviation exactly
  2^32, so ``z = (sum - 6*(2^32-1)) / 2^32`` is a bell-shaped variate on (-6, 6) that is
  exactly representable in float64,
* ``x = float32(scale * z)`` is two correctly rounded IEEE operations.

Being counter-based, any shard ``[start, start+n)`` of a long vector can be produced
independently (one rank per GPU each makes its own shard).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_MASK32 = np.uint64(0xFFFFFFFF)
_CHUNK = 1 << 20


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 output function on an array of uint64 counters (wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        z = z + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def irwin_hall12(n: int, seed: int, start: int = 0) -> np.ndarray:
    """float64 array of n bell-shaped variates (mean 0, std 1, support (-6, 6))."""
    out = np.empty(n, dtype=np.float64)
    seed_mix = _mix64(np.array([seed], dtype=np.uint64))[0]
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        idx = np.arange(start + lo, start + hi, dtype=np.uint64)
        acc = np.zeros(hi - lo, dtype=np.uint64)
        with np.errstate(over="ignore"):
            base = idx * np.uint64(6) + seed_mix
            for k in range(6):
                h = _mix64(base + np.uint64(k))
                acc += (h & _MASK32) + (h >> np.uint64(32))
        centred = acc.astype(np.int64) - np.int64(6 * ((1 << 32) - 1))
        out[lo:hi] = centred.astype(np.float64) / float(1 << 32)
    return out


def weights(shape, seed: int, scale: float = 0.05, start: int = 0) -> np.ndarray:
    """float32 tensor of the given shape, ``float32(scale * z)``; C order."""
    n = int(np.prod(shape))
    z = irwin_hall12(n, seed, start)
    return (z * float(scale)).astype(np.float32).reshape(shape)


# Shapes of the reference's two networks (neural_networks/le_net_300_100.py:19-25,
# neural_networks/le_net_5.py:20-44): (name, weight shape, bias shape).
LENET_300_100 = [
    ("dense1", (784, 300), (300,)),
    ("dense2", (300, 100), (100,)),
    ("out", (100, 10), (10,)),
]
LENET_5 = [
    ("conv1", (5, 5, 1, 20), (20,)),
    ("conv2", (5, 5, 20, 50), (50,)),
    ("dense1", (2450, 256), (256,)),
    ("out", (256, 10), (10,)),
]


def gpt2_small_layers():
    """(name, shape) of the 122 tensors of BASELINE configs[4] (a GPT-2-small-sized model, 124.4 M weights): token and
    position embeddings, then per block the four weight matrices, their biases and the two layer-norm gains."""
    shapes = [("wte", (50257, 768)), ("wpe", (1024, 768))]
    for l in range(12):
        shapes += [(f"h{l}.attn.c_attn", (768, 2304)), (f"h{l}.attn.c_proj", (768, 768)),
                   (f"h{l}.mlp.c_fc", (768, 3072)), (f"h{l}.mlp.c_proj", (3072, 768)),
                   (f"h{l}.b_attn", (2304,)), (f"h{l}.b_proj", (768,)), (f"h{l}.b_fc", (3072,)), (f"h{l}.b_proj2", (768,)),
                   (f"h{l}.ln1", (768,)), (f"h{l}.ln2", (768,))]
    return shapes
