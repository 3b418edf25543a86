"""CPU oracle for the prune -> k-means -> index/Huffman path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product (neural_network_compression_amd/) never does.

It restates, function by function, the reference's

    prune_weigth            /root/reference/neural_network_compression/common/utility.py:134-163
    get_weight_distribution /root/reference/neural_network_compression/common/utility.py:334-392
    get_quantized_weight    /root/reference/neural_network_compression/common/utility.py:172-240

with the heavy per-sample arithmetic in plain C (oracle/nnc_oracle.c) and the small
host-side steps in numpy.  The k-means arithmetic belongs to scikit-learn (pinned
~0.24.0 in /root/reference/pyproject.toml:13, 1.7.2 installed; absent from the reference
tree): its published Lloyd algorithm (sklearn/cluster/_kmeans.py:624-752, 1427-1554;
_k_means_lloyd.pyx:23-218; _k_means_common.pyx:167-311) is restated in ``kmeans_lloyd``.

Two accumulation modes for the M-step:
  * ``accum="A"``: float32 running sums in sample order == scikit-learn on ONE thread,
    bit for bit.  This is the mode pinned against tests/golden/ref_goldens.* (reference
    outputs generated in the build container by tests/golden/make_goldens.py).
  * ``accum="B"``: exact int64 sums of fixed-point images (see ``fix_shift``), centre =
    float32(ldexp(sum/count, -S)).  Order independent; this is what the HIP kernels
    compute, so GPU == mode B bit for bit at any GPU count.  The A<->B gap (about 1e-4
    relative on centres, scikit-learn's own float32 summation error) is measured in
    tests/test_oracle.py and reported in DESIGN.md.

Parity status: pinned (goldens from the reference itself) for prune, CDF, init and
k-means mode A.  Huffman code lengths have no reference implementation (the reference
never implemented Huffman coding): "parity unpinned" for ``huffman_lengths``.
"""
from __future__ import annotations

import ctypes
import heapq
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnnc_oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nnc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.orc_np_sum_f32.restype = ctypes.c_float
        L.orc_np_sum_f32.argtypes = [_f32p, ctypes.c_int64]
        L.orc_np_mean_f32.restype = ctypes.c_float
        L.orc_np_mean_f32.argtypes = [_f32p, ctypes.c_int64]
        L.orc_np_var_f32.restype = ctypes.c_float
        L.orc_np_var_f32.argtypes = [_f32p, ctypes.c_int64]
        L.orc_np_std_f32.restype = ctypes.c_float
        L.orc_np_std_f32.argtypes = [_f32p, ctypes.c_int64]
        L.orc_np_chunk_sums_f32.restype = None
        L.orc_np_chunk_sums_f32.argtypes = [_f32p, ctypes.c_int64, _f32p]
        L.orc_np_chunk_sqdev_f32.restype = None
        L.orc_np_chunk_sqdev_f32.argtypes = [_f32p, ctypes.c_int64, ctypes.c_float, _f32p]
        L.orc_fold_f32.restype = ctypes.c_float
        L.orc_fold_f32.argtypes = [_f32p, ctypes.c_int64]
        L.orc_prune_f32.restype = ctypes.c_int64
        L.orc_prune_f32.argtypes = [_f32p, ctypes.c_int64, ctypes.c_float, ctypes.c_int, _u8p, _f32p, _f32p]
        L.orc_apply_mask_f32.restype = None
        L.orc_apply_mask_f32.argtypes = [_f32p, _u8p, ctypes.c_int64]
        L.orc_minmax_f32.restype = None
        L.orc_minmax_f32.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _f32p, _i64p]
        L.orc_hist31_f32.restype = None
        L.orc_hist31_f32.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _i64p]
        L.orc_center_f32.restype = None
        L.orc_center_f32.argtypes = [_f32p, ctypes.c_int64, ctypes.c_float, _f32p]
        L.orc_estep_f32.restype = None
        L.orc_estep_f32.argtypes = [_f32p, ctypes.c_int64, _f32p, ctypes.c_int, _i32p]
        L.orc_mstep_a_f32.restype = None
        L.orc_mstep_a_f32.argtypes = [_f32p, ctypes.c_int64, _i32p, ctypes.c_int, _f32p, _f32p]
        L.orc_mstep_b_f32.restype = None
        L.orc_mstep_b_f32.argtypes = [_f32p, ctypes.c_int64, _i32p, ctypes.c_int, ctypes.c_int, _i64p, _i64p]
        L.orc_fix_f32.restype = ctypes.c_int64
        L.orc_fix_f32.argtypes = [ctypes.c_float, ctypes.c_int]
        L.orc_center_from_fix.restype = ctypes.c_float
        L.orc_center_from_fix.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        L.orc_dist_own_f32.restype = None
        L.orc_dist_own_f32.argtypes = [_f32p, ctypes.c_int64, _f32p, _i32p, _f32p]
        L.orc_gather_f32.restype = None
        L.orc_gather_f32.argtypes = [_f32p, _i32p, ctypes.c_int64, _f32p]
        L.orc_bincount_i32.restype = None
        L.orc_bincount_i32.argtypes = [_i32p, ctypes.c_int64, ctypes.c_int, _i64p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------- numpy reductions
def np_sum(a) -> np.float32:
    a = _f32c(a).ravel()
    return np.float32(lib().orc_np_sum_f32(_p(a, _f32p), a.size))


def np_mean(a) -> np.float32:
    a = _f32c(a).ravel()
    return np.float32(lib().orc_np_mean_f32(_p(a, _f32p), a.size))


def np_var(a) -> np.float32:
    a = _f32c(a).ravel()
    return np.float32(lib().orc_np_var_f32(_p(a, _f32p), a.size))


def np_std(a) -> np.float32:
    a = _f32c(a).ravel()
    return np.float32(lib().orc_np_std_f32(_p(a, _f32p), a.size))


def chunk_sums(a) -> np.ndarray:
    a = _f32c(a).ravel()
    out = np.empty((a.size + 8191) // 8192, dtype=np.float32)
    lib().orc_np_chunk_sums_f32(_p(a, _f32p), a.size, _p(out, _f32p))
    return out


def chunk_sqdev(a, mean) -> np.ndarray:
    a = _f32c(a).ravel()
    out = np.empty((a.size + 8191) // 8192, dtype=np.float32)
    lib().orc_np_chunk_sqdev_f32(_p(a, _f32p), a.size, np.float32(mean), _p(out, _f32p))
    return out


def fold(chunks) -> np.float32:
    c = _f32c(chunks).ravel()
    return np.float32(lib().orc_fold_f32(_p(c, _f32p), c.size))


# --------------------------------------------------------------------------- prune
def prune_threshold_f32(sigma: np.float32, threshold, std_smooth: bool) -> np.float32:
    """The float32 number t such that ``abs(w) < (np.std(w)*threshold or threshold)`` is
    ``abs(w) < t`` for every float32 w (numpy 2 / NEP 50 promotion rules)."""
    thr = (sigma * threshold) if std_smooth else threshold
    if isinstance(thr, np.float64):
        # a strongly typed float64 scalar: the comparison happens in float64;
        # |w| < t64  <=>  |w| < roundup32(t64) for float32 w
        t32 = np.float32(thr)
        if np.float64(t32) < np.float64(thr):
            t32 = np.nextafter(t32, np.float32(np.inf), dtype=np.float32)
        return np.float32(t32)
    return np.float32(thr)


def prune_weigth(original_weigth: np.ndarray, threshold=0.25, std_smooth=True) -> np.ndarray:
    """utility.py:134-163: mask = |w| < (std(w)*threshold | threshold); w[mask] = 0 in place."""
    w = original_weigth
    assert w.dtype == np.float32 and w.flags.c_contiguous
    flat = w.reshape(-1)
    sigma = np_std(flat) if std_smooth else np.float32(0)
    thr = prune_threshold_f32(sigma, threshold, std_smooth)
    mask = np.empty(flat.size, dtype=np.uint8)
    lib().orc_prune_f32(_p(flat, _f32p), flat.size, thr, 0, _p(mask, _u8p), None, None)
    return mask.view(np.bool_).reshape(w.shape)


# --------------------------------------------------------------------------- CDF
def _interp_linear(x, y, x_new):
    """scipy.interpolate.interp1d(x, y, 'linear')(x_new) restated
    (scipy/interpolate/_interpolate.py, interp1d._call_linear)."""
    x = np.asarray(x)
    y = np.asarray(y, dtype=np.float64)
    x_new = np.asarray(x_new)
    idx = np.searchsorted(x, x_new)
    idx = idx.clip(1, len(x) - 1).astype(int)
    lo = idx - 1
    hi = idx
    x_lo = x[lo]
    x_hi = x[hi]
    y_lo = y[lo]
    y_hi = y[hi]
    slope = (y_hi - y_lo) / (x_hi - x_lo)[:]
    return slope * (x_new - x_lo)[:] + y_lo


def cdf_from_counts(steps: np.ndarray, counts) -> tuple:
    """utility.py:374-392 given the 32 steps and the 31 integer counts."""
    x = steps[:-1]
    tot_counter = np.array([int(c) for c in counts]) / (np.sum([int(c) for c in counts]))
    cdf = []
    for i in range(len(tot_counter)):
        if i == 0:
            cdf.append(tot_counter[i])
        else:
            cdf.append(tot_counter[i] + cdf[i - 1])
    cdf = np.array(cdf)
    cdf = cdf / cdf[-1]
    xnew = np.linspace(min(x), max(x), 300)
    return xnew, _interp_linear(x, cdf, xnew)


def get_weight_distribution(weight_matrix: np.ndarray):
    """utility.py:334-392."""
    w = _f32c(weight_matrix).ravel()
    mn = np.float32(0)
    mx = np.float32(0)
    mn_c = ctypes.c_float()
    mx_c = ctypes.c_float()
    cnt = ctypes.c_int64()
    if w.size == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    lib().orc_minmax_f32(_p(w, _f32p), w.size, 0, ctypes.byref(mn_c), ctypes.byref(mx_c), ctypes.byref(cnt))
    mn, mx = np.float32(mn_c.value), np.float32(mx_c.value)
    steps = np.linspace(mn, mx, num=32)
    assert steps.dtype == np.float32
    counts = np.zeros(31, dtype=np.int64)
    lib().orc_hist31_f32(_p(w, _f32p), w.size, 0, _p(steps, _f32p), _p(counts, _i64p))
    return cdf_from_counts(steps, counts)


# --------------------------------------------------------------------------- init
def init_space(layer_weight: np.ndarray, bits: int, mode: str, cdfs=None) -> np.ndarray:
    """utility.py:206-226 (the three explicit-init modes)."""
    if mode == "linear":
        min_ = layer_weight.min()
        max_ = layer_weight.max()
        return np.linspace(min_, max_, num=2 ** bits)
    if mode == "density" and cdfs is not None:
        tmp = np.linspace(0, 1, num=(2 ** bits) + 1)
        xval, yval = cdfs[0], cdfs[1]
        space = []
        for i in range(len(tmp)):
            minval = min(yval, key=lambda x: abs(x - tmp[i]))
            idx_val = np.argmax(yval == minval)
            space.append(xval[idx_val])
        return np.array(space)
    if mode == "forgy":
        flat = layer_weight.flatten()
        return np.random.choice(flat, size=2 ** bits)
    raise Exception(" error mode not found")


# --------------------------------------------------------------------------- fixed point (mode B)
def fix_shift(absmax: float, n_total: int) -> int:
    """Shift S of the fixed-point image q = rint(v * 2^S): the largest S with
    |q| <= 2^min(28, 62-L) for every |v| <= absmax, L = ceil(log2(n_total)): one image is a
    32-bit integer, four of them add up inside int32 (the device adds a float4 at a time), and a
    sum over n_total images cannot overflow int64.  (Same rule in the product host code.)"""
    L = max(1, (int(n_total) - 1).bit_length())
    if not (absmax > 0) or not math.isfinite(absmax):
        return 0
    _, P = math.frexp(float(absmax))  # absmax = m * 2^P, m in [0.5, 1)  =>  |v| < 2^P
    return min(28, 62 - L) - P


def fix(v, S: int) -> int:
    return int(lib().orc_fix_f32(np.float32(v), int(S)))


# --------------------------------------------------------------------------- Lloyd
class KMeansResult:
    """Carries the three attributes the reference reads from the fitted sklearn model."""

    def __init__(self, centers, labels, n_iter, trace=None, strict=False):
        self.cluster_centers_ = np.asarray(centers, dtype=np.float32).reshape(-1, 1)
        self.labels_ = np.asarray(labels, dtype=np.int32)
        self.n_iter_ = int(n_iter)
        self.trace = trace
        self.strict = strict


def estep(xc: np.ndarray, c: np.ndarray) -> np.ndarray:
    xc = _f32c(xc).ravel()
    c = _f32c(c).ravel()
    labels = np.empty(xc.size, dtype=np.int32)
    lib().orc_estep_f32(_p(xc, _f32p), xc.size, _p(c, _f32p), c.size, _p(labels, _i32p))
    return labels


def lloyd_iter(xc, c_old, accum="A", S=0, want_labels=True, reloc=None, info=None):
    """One sklearn lloyd_iter_chunked_dense(update_centers=True) on centred data.
    Returns (labels, centers_new float32, counts, shift float32[K], n_empty)."""
    L = lib()
    K = c_old.size
    labels = estep(xc, c_old)
    if accum == "A":
        sums = np.empty(K, dtype=np.float32)
        wic = np.empty(K, dtype=np.float32)
        L.orc_mstep_a_f32(_p(xc, _f32p), xc.size, _p(labels, _i32p), K, _p(sums, _f32p), _p(wic, _f32p))
    else:
        sums = np.empty(K, dtype=np.int64)
        wic = np.empty(K, dtype=np.int64)
        L.orc_mstep_b_f32(_p(xc, _f32p), xc.size, _p(labels, _i32p), K, S, _p(sums, _i64p), _p(wic, _i64p))
    # ---- _relocate_empty_clusters_dense (_k_means_common.pyx:167-211)
    empty = np.where(wic == 0)[0].astype(np.int32)
    n_empty = int(empty.size)
    if n_empty:
        d = np.empty(xc.size, dtype=np.float32)
        L.orc_dist_own_f32(_p(xc, _f32p), xc.size, _p(c_old, _f32p), _p(labels, _i32p), _p(d, _f32p))
        if reloc is None:
            reloc = "argpartition" if accum == "A" else "descending"
        if info is not None and np.max(d) != 0:   # (all distances zero: scikit-learn relocates nothing)
            # an exact tie at the selection cut: the samples sklearn takes then depend on numpy's introselect
            ds = np.sort(d)
            info["reloc_events"] = info.get("reloc_events", 0) + 1
            info["reloc_multi"] = info.get("reloc_multi", 0) + (n_empty > 1)
            if n_empty < d.size and ds[-n_empty] == ds[-n_empty - 1] and ds[-n_empty] != 0:
                info["reloc_ties"] = info.get("reloc_ties", 0) + 1
        if reloc == "argpartition":
            # scikit-learn: whatever order numpy.argpartition leaves the top n_empty indices in
            far = np.argpartition(d, -n_empty)[: -n_empty - 1 : -1].astype(np.int32)
        else:
            # mode B (the device rule): descending distance, equal distances by descending value
            # (samples equal in both are interchangeable).  Same set of samples up to ties at the
            # cut; same pairing as scikit-learn whenever n_empty == 1.
            far = np.lexsort((np.arange(d.size), xc, d))[::-1][:n_empty].astype(np.int32)
        if np.max(d) != 0:
            for idx in range(n_empty):
                new, far_idx = int(empty[idx]), int(far[idx])
                old = int(labels[far_idx])
                if accum == "A":
                    v = np.float32(xc[far_idx] * np.float32(1.0))
                    sums[old] = np.float32(sums[old] - v)
                    sums[new] = v
                    wic[new] = np.float32(1.0)
                    wic[old] = np.float32(wic[old] - np.float32(1.0))
                else:
                    v = fix(xc[far_idx], S)
                    sums[old] -= v
                    sums[new] = v
                    wic[new] = 1
                    wic[old] -= 1
    # ---- _average_centers (_k_means_common.pyx:274-296), in place and in index order
    argmax_w = int(np.argmax(wic))
    if accum == "A":
        cen = sums.copy()
        for j in range(K):
            if wic[j] > 0:
                alpha = np.float32(1.0 / float(wic[j]))
                cen[j] = np.float32(cen[j] * alpha)
            else:
                cen[j] = cen[argmax_w]
    else:
        cen = np.empty(K, dtype=np.float32)
        for j in range(K):
            if wic[j] > 0:
                cen[j] = np.float32(L.orc_center_from_fix(int(sums[j]), int(wic[j]), S))
            elif argmax_w < j:
                cen[j] = cen[argmax_w]  # biggest cluster, already averaged
            else:
                cen[j] = np.float32(math.ldexp(float(int(sums[argmax_w])), -S))  # its raw sum (sklearn quirk)
    # ---- _center_shift (_k_means_common.pyx:298-311): sqrt((new-old)^2) in float32
    t = (cen - c_old).astype(np.float32)
    shift = np.sqrt((t * t).astype(np.float32)).astype(np.float32)
    return labels, cen, wic.copy(), shift, n_empty


DEVICE_REF_NMAX, DEVICE_REF_KMAX = 4096, 128


def device_arith(n, k):
    """(accum, reloc) of the product's single-GPU fit of n weights with k centres: short tensors are fitted in the
    reference's own arithmetic (mode A sums) with the device's relocation order, everything else in mode B."""
    return ("A", "descending") if (n <= DEVICE_REF_NMAX and k <= DEVICE_REF_KMAX) else ("B", "descending")


def kmeans_lloyd(x, init, accum="A", max_iter=300, tol=1e-4, n_total=None, keep_trace=False, reloc=None):
    """KMeans(n_clusters=K, init=init[:,None], n_init=1, algorithm='full').fit(x[:,None])
    (utility.py:237-238) -> KMeansResult.  accum="device": what device_arith says for this size."""
    x = _f32c(x).ravel()
    n = x.size
    if accum == "device":
        accum, reloc = device_arith(n, np.asarray(init).size)
    L = lib()
    # _tolerance (_kmeans.py:279-287): np.mean(np.var(X, axis=0)) * tol, all float32
    tol_ = np.float32(np_var(x) * np.float32(tol))
    x_mean = np_mean(x)
    xc = np.empty_like(x)
    L.orc_center_f32(_p(x, _f32p), n, x_mean, _p(xc, _f32p))
    centers = (np.asarray(init, dtype=np.float32).ravel() - x_mean).astype(np.float32)
    K = centers.size
    S = 0
    if accum == "B":
        S = fix_shift(float(np.max(np.abs(xc))) if n else 0.0, n if n_total is None else n_total)
    labels_old = np.full(n, -1, dtype=np.int32)
    strict = False
    trace = [] if keep_trace else None
    info = {}
    n_iter = 0
    for i in range(max_iter):
        labels, centers_new, counts, shift, n_empty = lloyd_iter(xc, centers, accum, S, reloc=reloc, info=info)
        centers = centers_new
        n_iter = i + 1
        tot = np.float32((shift ** 2).sum())
        if keep_trace:
            trace.append({"centers": centers.copy(), "counts": counts.astype(np.int64), "shift_tot": tot,
                          "label_counts": np.bincount(labels, minlength=K).astype(np.int64),
                          "labels": labels if keep_trace == "labels" else None, "n_empty": n_empty})
        if np.array_equal(labels, labels_old):
            strict = True
            break
        elif tot <= tol_:
            break
        labels_old = labels
    if not strict:
        labels = estep(xc, centers)
    final = (centers + x_mean).astype(np.float32)
    res = KMeansResult(final, labels, n_iter, trace, strict)
    res.tol_ = tol_
    res.x_mean_ = x_mean
    res.fix_shift_ = S
    res.centers_centred_ = centers
    res.reloc_info_ = info
    return res


def get_quantized_weight(layer_weight, bits=4, mode="linear", cdfs=None, accum="A", reloc=None):
    """utility.py:172-240 for the three explicit-init modes."""
    if np.prod(layer_weight.shape) < (2 ** bits) + 1:
        print("not enough bits:", np.prod(layer_weight.shape), " vs ", 2 ** bits)
        return layer_weight, None
    space = init_space(layer_weight, bits, mode, cdfs)
    km = kmeans_lloyd(layer_weight.reshape(-1), space, accum=accum, reloc=reloc)  # accum="device": device_arith
    km.init_space_ = np.asarray(space)
    ris = km.cluster_centers_[km.labels_].reshape(layer_weight.shape)
    return ris, km


# --------------------------------------------------------------------------- Huffman (unpinned)
def huffman_lengths(counts):
    """Code length per symbol for a Huffman code over the symbols with count > 0.

    NO reference implementation exists (the reference never wrote Huffman coding; the
    spec is Deep Compression section 4) -- parity unpinned.  Definition used by both this
    oracle and the product: min-heap keyed by (weight, smallest symbol in the subtree);
    pop two, push (w1+w2, min(s1,s2)); a symbol's length is its depth; zero-count symbols
    get length 0; a single used symbol gets length 1.
    Returns (lengths uint8[K], hist int64[max_len+1], total_bits int)."""
    counts = [int(c) for c in counts]
    K = len(counts)
    lengths = np.zeros(K, dtype=np.uint8)
    heap = [(c, s, [s]) for s, c in enumerate(counts) if c > 0]
    if len(heap) == 1:
        lengths[heap[0][1]] = 1
    else:
        heapq.heapify(heap)
        while len(heap) > 1:
            w1, s1, m1 = heapq.heappop(heap)
            w2, s2, m2 = heapq.heappop(heap)
            for s in m1 + m2:
                lengths[s] += 1
            heapq.heappush(heap, (w1 + w2, min(s1, s2), m1 + m2))
    hist = np.bincount(lengths, minlength=1).astype(np.int64)
    total = int(sum(int(lengths[s]) * counts[s] for s in range(K)))
    return lengths, hist, total


# --------------------------------------------------------------------------- kmeans++ mode (utility.py:228-232)
def _pp_dist(c, xc):
    """sklearn's _euclidean_distances(c[None, :], X, squared=True) for float32 inputs with one feature
    (metrics/pairwise.py: float32 data are upcast chunk-wise): ((-2 * (c * x)) + c * c) + x * x in float64, cast to
    float32, clipped at 0."""
    c64 = np.float64(c)
    x64 = xc.astype(np.float64)
    d = -2.0 * (c64 * x64)
    d += c64 * c64
    d += x64 * x64
    d32 = d.astype(np.float32)
    np.maximum(d32, 0, out=d32)
    return d32


PP_BLOCK, PP_GROUP = 1024, 256


def _pp_block_sums(d32: np.ndarray) -> np.ndarray:
    """float64 block sums of float32 values in the order csrc/nnc_pp.hip fixes: blocks of 1024 samples; lane l adds the
    elements {256 t + 4 l + u : t, u = 0..3} of its block in that order; the 64 lane sums meet in an xor butterfly."""
    n = d32.size
    nblk = (n + PP_BLOCK - 1) // PP_BLOCK
    pad = np.zeros(nblk * PP_BLOCK, dtype=np.float64)   # (+0.0 for the missing tail: adding it changes nothing)
    pad[:n] = d32
    v = pad.reshape(nblk, 4, 64, 4).transpose(0, 2, 1, 3).reshape(nblk, 64, 16)
    acc = np.zeros((nblk, 64), dtype=np.float64)
    for s in range(16):
        acc = acc + v[:, :, s]
    lanes = np.arange(64)
    for off in (32, 16, 8, 4, 2, 1):
        acc = acc + acc[:, lanes ^ off]
    return acc[:, 0].copy()


def _pp_scan(S: np.ndarray):
    """W[b] = left-to-right sum of the blocks of b's group before b; G[g] = left-to-right sum of the group totals before g
    (G[-1] = grand total)."""
    nblk = S.size
    ngroups = (nblk + PP_GROUP - 1) // PP_GROUP
    W = np.zeros(nblk, dtype=np.float64)
    G = np.zeros(ngroups + 1, dtype=np.float64)
    for g in range(ngroups):
        acc = np.float64(0.0)
        for b in range(g * PP_GROUP, min(nblk, (g + 1) * PP_GROUP)):
            W[b] = acc
            acc = acc + S[b]
        G[g + 1] = G[g] + acc
    return W, G


def _pp_search(closest32, S, W, G, r):
    """searchsorted(running sum, r) in the three-level order of csrc/nnc_pp.hip (k_pp_select)."""
    n, nblk = closest32.size, S.size
    ngroups = G.size - 1
    hit = np.nonzero(G[1:] >= r)[0]
    g = int(hit[0]) if hit.size else ngroups - 1
    b0, b1 = g * PP_GROUP, min(nblk, (g + 1) * PP_GROUP)
    hit = np.nonzero(G[g] + (W[b0:b1] + S[b0:b1]) >= r)[0]
    b = b0 + (int(hit[0]) if hit.size else b1 - b0 - 1)
    base = G[g] + W[b]
    i0, i1 = b * PP_BLOCK, min(n, (b + 1) * PP_BLOCK)
    run = np.cumsum(closest32[i0:i1], dtype=np.float64)      # sequential float64 running sum
    hit = np.nonzero(base + run >= r)[0]
    return i0 + (int(hit[0]) if hit.size else i1 - i0 - 1)


def kmeans_plusplus(xc: np.ndarray, K: int, random_state=None) -> tuple:
    """sklearn.cluster._kmeans._kmeans_plusplus (cluster/_kmeans.py:163-253 in 1.7.2) on centred float32 data with one
    feature and unit weights, drawing from the same generator in the same order.  Two orders of summation are pinned
    down here that scikit-learn leaves open or sequential: the potentials (``closest_dist_sq @ sample_weight``, a float32
    BLAS GEMV whose order is the BLAS kernel's) and the running sum the candidates are looked up in (np.cumsum in
    float64) are float64 sums in the blocked order of csrc/nnc_pp.hip.  The potential differs from scikit-learn's in
    its last bits, so on long vectors a drawn candidate can come out as a neighbouring sample.
    Returns (centres float32[K], indices)."""
    rs = np.random.mtrand._rand if random_state is None else random_state
    xc = _f32c(xc).ravel()
    n = xc.size
    n_local_trials = 2 + int(np.log(K))
    centers = np.empty(K, dtype=np.float32)
    indices = np.full(K, -1, dtype=np.int64)
    # random_state.choice(n, p=w / w.sum()): one uniform double, inverted through the float64 cdf of n equal float32
    # probabilities p0 = float32(1) / float32(n); its partial sums j * p0 are exact below 2^29 samples, so
    # cdf[j - 1] = fl64(j * p0 / (n * p0)) and the index is #{j : cdf[j - 1] <= u}  (numpy: searchsorted(side="right"))
    u = rs.random_sample()
    p0 = np.float64(np.float32(1.0) / np.float32(n))
    tot = np.float64(n) * p0
    j = max(0, min(n, int(u * n)))
    while j < n and (np.float64(j + 1) * p0) / tot <= u:
        j += 1
    while j > 0 and (np.float64(j) * p0) / tot > u:
        j -= 1
    center_id = min(j, n - 1)
    centers[0] = xc[center_id]
    indices[0] = center_id
    closest = _pp_dist(centers[0], xc)
    S = _pp_block_sums(closest)
    W, G = _pp_scan(S)
    pot = np.float32(G[-1])
    for c in range(1, K):
        rand_vals = rs.uniform(size=n_local_trials) * np.float64(pot)
        cand = [_pp_search(closest, S, W, G, r) for r in rand_vals]
        best = None
        for t in range(n_local_trials):
            d = np.minimum(closest, _pp_dist(xc[cand[t]], xc))
            St = _pp_block_sums(d)
            Wt, Gt = _pp_scan(St)
            p = np.float32(Gt[-1])
            if best is None or p < best[0]:
                best = (p, t, d, St, Wt, Gt)
        pot, t, closest, S, W, G = best
        centers[c] = xc[cand[t]]
        indices[c] = cand[t]
    return centers, indices


def kmeans_plusplus_fit(x, K, accum="B", random_state=None):
    """KMeans(n_clusters=K).fit(x[:, None]) as utility.py:229-230 calls it (scikit-learn 1.7: init="k-means++",
    n_init="auto" -> 1, max_iter=300, tol=1e-4, the global NumPy generator): mean-centre, seed, Lloyd."""
    x = _f32c(x).ravel()
    x_mean = np_mean(x)
    xc = np.empty_like(x)
    lib().orc_center_f32(_p(x, _f32p), x.size, x_mean, _p(xc, _f32p))
    seeds, idx = kmeans_plusplus(xc, K, random_state)
    # kmeans_lloyd subtracts the mean from its init in float32; hand it seeds + mean so that ... no: hand it the samples
    # themselves (x[idx] - mean == xc[idx] bit for bit, the same float32 subtraction)
    km = kmeans_lloyd(x, x[idx], accum=accum)
    km.seed_indices_ = idx
    return km
