"""Drop-in counterparts of the reference's compression primitives, running on MI355X.

Same names, arguments, return values and error behaviour as
``neural_network_compression/common/utility.py`` of the reference:

    prune_weigth(original_weigth, threshold=0.25, std_smooth=True)          utility.py:134-163
    get_quantized_weight(layer_weight, bits=4, mode="linear", cdfs=None)     utility.py:172-240
    get_weight_distribution(weight_matrix)                                   utility.py:334-392

Arguments may be NumPy arrays (the reference's calling convention: the data crosses PCIe,
results come back as NumPy arrays, ``prune_weigth`` still mutates its argument) or float32
CUDA tensors already resident in HBM (no host copy of the weights at all; results are CUDA
tensors).  The arithmetic runs in the HIP kernels of csrc/nnc_hip.hip through the C ABI of
include/nnc.h; the few K-sized host steps (linspace, the density pick, the forgy draw) use
the same NumPy calls the reference makes so that their dtype/rounding behaviour is inherited.
There is no CPU fallback: without the HIP library these functions raise.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import kmeans as _kmeans
from .. import ops

_DEVICE = None


def default_device() -> torch.device:
    global _DEVICE
    if _DEVICE is None:
        if not torch.cuda.is_available():
            raise RuntimeError("neural_network_compression_amd needs an MI355X (no GPU visible, no CPU fallback)")
        _DEVICE = torch.device("cuda", torch.cuda.current_device())
    return _DEVICE


def _to_device(a):
    """-> (flat float32 CUDA tensor, was_numpy).  NumPy input is copied host->HBM."""
    if isinstance(a, torch.Tensor):
        if not a.is_cuda:
            raise TypeError("torch tensors must live on the GPU; pass a NumPy array for host data")
        if a.dtype != torch.float32:
            raise TypeError("weights must be float32")
        return a.contiguous().reshape(-1), False
    arr = np.asarray(a)
    if arr.dtype != np.float32:
        raise TypeError(f"weights must be float32 (the reference feeds Keras float32 weights), got {arr.dtype}")
    t = torch.from_numpy(np.ascontiguousarray(arr).reshape(-1)).to(default_device())
    return t, True


def _threshold_f32(sigma, threshold, std_smooth):
    """float32 t with (|w| < t) == (|w| < np.std(w)*threshold  or  |w| < threshold) for float32 w
    under NumPy-2 promotion: python scalars are weak (float32 arithmetic), a float64 NumPy
    scalar forces a float64 comparison, which equals comparing against t rounded UP to float32."""
    thr = (np.float32(sigma) * threshold) if std_smooth else threshold
    if isinstance(thr, np.float64):
        t32 = np.float32(thr)
        if np.float64(t32) < thr:
            t32 = np.nextafter(t32, np.float32(np.inf), dtype=np.float32)
        return np.float32(t32)
    return np.float32(thr)


# ------------------------------------------------------------------------------------------
def prune_weigth(original_weigth, threshold=0.25, std_smooth=True):
    """Zero the weights whose magnitude is below ``threshold`` (times the tensor's standard
    deviation if ``std_smooth``) IN PLACE and return the boolean mask of the zeroed entries.
    Reference: utility.py:134-163 (np.std, np.abs(w) < thr, w[mask] = 0)."""
    x, was_numpy = _to_device(original_weigth)
    shape = tuple(original_weigth.shape)
    simple = isinstance(threshold, (int, float, np.float32)) and not isinstance(threshold, (np.float64, bool))
    if x.numel() == 0:
        mask = torch.zeros(0, dtype=torch.uint8, device=x.device)
    elif simple:
        mask, _, _ = ops.prune_(x, np.float32(threshold), bool(std_smooth))
    else:
        sigma = ops.moments(x)[2].cpu().numpy()[0] if std_smooth else np.float32(0)
        thr = torch.tensor([_threshold_f32(sigma, threshold, std_smooth)], dtype=torch.float32, device=x.device)
        mask, _ = ops.threshold_mask_(x, thr)
    if was_numpy:
        np.copyto(original_weigth, x.cpu().numpy().reshape(shape))
        return mask.cpu().numpy().view(np.bool_).reshape(shape)
    if x.data_ptr() != original_weigth.data_ptr():
        original_weigth.copy_(x.view(shape))
    return mask.view(torch.bool).view(shape)


# ------------------------------------------------------------------------------------------
def _cdf_from_counts(steps: np.ndarray, counts: np.ndarray):
    """utility.py:374-392 on the 32 steps and the 31 integer bin counts (host, 31 values): normalised counts, their running
    float64 sum, divided by its last entry, then scipy.interpolate.interp1d(x, cdf, "linear") at 300 points -- its
    ``_call_linear`` spelled out with the same dtypes (x and the 300 points float32, the cdf float64), which gives the same
    bits without building the interpolator object (tests/test_abi.py::test_cdf_from_counts_is_scipys)."""
    x = steps[:-1]
    c = np.asarray(counts, dtype=np.int64)
    tot_counter = c / np.sum(c)          # int / int -> float64
    cdf = np.cumsum(tot_counter)         # sequential float64 adds, as the reference's python loop
    cdf = cdf / cdf[-1]
    xnew = np.linspace(x.min(), x.max(), 300)
    idx = np.searchsorted(x, xnew).clip(1, len(x) - 1).astype(int)
    lo, hi = idx - 1, idx
    x_lo, y_lo = x[lo], cdf[lo]
    slope = (cdf[hi] - y_lo) / (x[hi] - x_lo)
    return xnew, slope * (xnew - x_lo) + y_lo


def _weight_distribution_device(x: torch.Tensor, skip_zeros: bool):
    mm, cnt = ops.minmax(x, skip_zeros=skip_zeros)
    host = mm.cpu().numpy()
    if skip_zeros and int(cnt.item()) == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    steps = np.linspace(np.float32(host[0]), np.float32(host[1]), num=32)  # float32 under NumPy 2
    steps_d = torch.from_numpy(np.ascontiguousarray(steps, dtype=np.float32)).to(x.device)
    counts = ops.hist31(x, steps_d, skip_zeros=skip_zeros).cpu().numpy()
    return _cdf_from_counts(steps, counts)


def get_weight_distribution(weight_matrix, skip_zeros: bool = False):
    """(xnew[300], cdf[300]): 31-bin histogram over linspace(min, max, 32) -> cumulative ->
    normalised -> linear interpolation at 300 points.  Reference: utility.py:334-392.

    ``skip_zeros=True`` (extension) ignores exact zeros on the device instead of having the
    caller strip them first (Trainer.quantize does ``numpy.delete`` of the zeros,
    common/trainer.py:55-59); the result is identical to passing the stripped vector."""
    if not isinstance(weight_matrix, torch.Tensor) and np.asarray(weight_matrix).size == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    x, _ = _to_device(weight_matrix)
    return _weight_distribution_device(x, skip_zeros)


# ------------------------------------------------------------------------------------------
def _init_space(x: torch.Tensor, n: int, bits: int, mode: str, cdfs):
    """The reference's three explicit centroid initialisations (utility.py:206-226)."""
    if mode == "linear":
        mm, _ = ops.minmax(x)
        host = mm.cpu().numpy()
        return np.linspace(np.float32(host[0]), np.float32(host[1]), num=2 ** bits)
    if mode == "density" and cdfs is not None:
        # for each target t in linspace(0, 1, 2**bits + 1): the x of the FIRST cdf value closest to t.
        # The reference does this with python's min(key=abs(y - t)) and argmax(y == min)
        # (utility.py:212-221); argmin over the same float64 |y - t| picks the same first minimum.
        tmp = np.linspace(0, 1, num=(2 ** bits) + 1)
        xval, yval = np.asarray(cdfs[0]), np.asarray(cdfs[1], dtype=np.float64)
        idx = np.abs(yval[None, :] - tmp[:, None]).argmin(axis=1)
        return xval[idx]
    if mode == "forgy":
        # np.random.choice(flat, size=K) draws K indices with the legacy global RNG
        # (randint(0, N, K)) and gathers; draw the same indices, gather on the device
        idx = np.random.randint(0, n, size=2 ** bits)
        return x[torch.from_numpy(idx).to(x.device)].cpu().numpy()
    raise Exception(" error mode not found")


def _first_seed_index(u: float, n: int) -> int:
    """random_state.choice(n, p=w / w.sum()) for n unit weights, given its one uniform draw u (numpy mtrand.choice: the
    float64 cdf of the probabilities, normalised by its last entry, searched with side="right").  The probabilities are
    n copies of p0 = float32(1) / float32(n); their partial sums j * p0 are exact in float64 below 2^29 samples, so
    cdf[j - 1] = fl64(j * p0 / (n * p0)) and the index is the number of entries <= u."""
    p0 = np.float64(np.float32(1.0) / np.float32(n))
    tot = np.float64(n) * p0
    j = max(0, min(n, int(u * n)))
    while j < n and (np.float64(j + 1) * p0) / tot <= u:
        j += 1
    while j > 0 and (np.float64(j) * p0) / tot > u:
        j -= 1
    return min(j, n - 1)


def kmeans_plusplus_init(x: torch.Tensor, k: int, stats=None):
    """scikit-learn's k-means++ seeds for the flattened float32 CUDA vector x, drawn from NumPy's GLOBAL generator
    exactly as ``KMeans(n_clusters=k).fit`` would draw them (cluster/_kmeans.py:163-253, reached from the reference's
    utility.py:229-230).  Returns (seeds float32[k] = x[indices], indices int64[k]); see include/nnc.h
    (nnc_kmeanspp_seed_f32) for what is and is not reproducible of scikit-learn's float32 BLAS potential."""
    import ctypes

    from .. import _native as nat

    L = nat.load()
    n = x.numel()
    if n < k:
        raise ValueError(f"n_samples={n} should be >= n_clusters={k}.")
    if n >= 1 << 29:
        raise ValueError("kmeans++ seeding supports fewer than 2^29 samples")
    if stats is None:
        stats = _kmeans.LayerStats(x)
    trials = int(L.nnc_kmeanspp_trials(k))
    u0 = np.random.random_sample()
    first = _first_seed_index(u0, n)
    uni = np.random.uniform(size=(k - 1) * trials) if k > 1 else np.zeros(0)   # uniform(size=trials) per round, in order
    uni_d = torch.from_numpy(np.ascontiguousarray(uni, dtype=np.float64)).to(x.device) if k > 1 else None
    seeds = torch.empty(k, dtype=torch.float32, device=x.device)
    ids = torch.empty(k, dtype=torch.int64, device=x.device)
    ws_bytes = int(L.nnc_kmeanspp_workspace_bytes(n, k))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    nat.check(L.nnc_kmeanspp_seed_f32(x.data_ptr(), n, float(stats.mean), k, first, ops._ptr(uni_d), seeds.data_ptr(), ids.data_ptr(),
                                      ws.data_ptr(), ws_bytes, ops._stream(x)))
    return x[ids].cpu().numpy(), ids.cpu().numpy()


def get_quantized_weight(layer_weight, bits=4, mode="linear", cdfs=None, group=None, arith="auto", reloc="auto"):
    """Replace every weight by the centroid of its k-means cluster (2**bits centroids;
    2**bits + 1 for ``density``).  Returns ``(quantized weights, fitted model)`` where the
    model exposes ``cluster_centers_``, ``labels_`` and ``n_iter_`` like the scikit-learn
    object the reference returns.  Reference: utility.py:172-240.

    Same corner cases: fewer than ``2**bits + 1`` weights -> prints "not enough bits" and
    returns ``(layer_weight, None)``; unknown mode, or ``density`` without ``cdfs`` ->
    ``Exception(" error mode not found")``.  ``kmeans++`` (utility.py:228-232) seeds with scikit-learn's
    k-means++ from NumPy's global generator (kmeans_plusplus_init), then runs the same Lloyd fit.

    ``group``: a torch.distributed process group when ``layer_weight`` is this rank's
    contiguous shard of a longer vector (shards start on multiples of 8192 elements).
    ``arith``: "auto" fits tensors of up to 4096 weights in scikit-learn's own summation order (kmeans.fit_reference: the
    reference's centres bit for bit) and longer ones with exact fixed-point sums; "fixed" forces the latter.
    ``reloc``: "auto" re-seeds empty clusters on the device (descending distance, ties by descending value); "reference" lets
    ``numpy.argpartition`` pick the far samples from the distances in sample order, as scikit-learn does
    (_k_means_common.pyx:186-187): the reference's own pairing and its own choice at a tie, for one host read per event."""
    n = int(np.prod(layer_weight.shape))
    if group is None and n < (2 ** bits) + 1:
        print("not enough bits:", n, " vs ", 2 ** bits)
        return layer_weight, None
    if mode not in ("linear", "forgy", "kmeans++") and not (mode == "density" and cdfs is not None):
        raise Exception(" error mode not found")
    x, was_numpy = _to_device(layer_weight)
    if mode == "kmeans++":
        # utility.py:228-232: KMeans(n_clusters=2 ** bits) with scikit-learn's defaults (k-means++ seeding from the global
        # NumPy generator, one run, max_iter 300, tol 1e-4)
        if group is not None:
            raise NotImplementedError("kmeans++ seeding needs the whole vector on one GPU")
        space, _ = kmeans_plusplus_init(x, 2 ** bits)
        model, values = _kmeans.fit_vector(x, space, want_values=True, arith=arith, **({"reloc": reloc} if reloc != "auto" else {}))
        shape = tuple(layer_weight.shape)
        return (values.cpu().numpy().reshape(shape), model) if was_numpy else (values.view(shape), model)
    if group is not None and mode != "density":
        raise NotImplementedError("sharded fits take an explicit init: use kmeans.DeviceKMeans")
    space = _init_space(x, n, bits, mode, cdfs)
    model, values = _kmeans.fit_vector(x, np.asarray(space, dtype=np.float32), want_values=True, arith=arith, group=group,
                                       **({"reloc": reloc} if reloc != "auto" else {}))
    shape = tuple(layer_weight.shape)
    if was_numpy:
        return values.cpu().numpy().reshape(shape), model
    return values.view(shape), model
