"""Lloyd k-means on a flattened weight vector resident in HBM.

Host-side driver of the device state machine in csrc/nnc_hip.hip (nnc_kmeans_*): it is
the counterpart of ``KMeans(n_clusters=K, init=space.reshape(-1,1), n_init=1,
algorithm="full").fit(w.reshape(-1,1))`` as the reference calls it
(neural_network_compression/common/utility.py:237-238), following scikit-learn's
``KMeans.fit`` / ``_kmeans_single_lloyd`` (sklearn/cluster/_kmeans.py:1427-1554, 624-752):

  tol = float32(var(X)) * float32(1e-4); X -= mean; init -= mean
  repeat <= 300 times: E-step, M-step (+ empty-cluster relocation), centre shift;
      stop if labels == previous labels, else if sum(shift^2) <= tol
  one more E-step unless the stop was the label test; centres += mean

Per-cluster sums are exact fixed-point integers (include/nnc.h), so the result does not
depend on block order or on how many GPUs share the vector.  With ``group`` (a
torch.distributed process group, one rank per GPU) every rank holds one contiguous shard
starting on a multiple of 8192 elements and the only data-path exchange is one small
all-reduce (2K int64) per iteration.
"""
from __future__ import annotations

import ctypes
import math
import threading
import time

import numpy as np
import torch

from . import _native as nat
from . import ops

MAX_ITER = 300  # scikit-learn default, which the reference does not override
TOL = 1e-4
SORT_MIN_WEIGHTS = 512  # from here on a fit runs on a value-sorted copy: even where the sort buys the streaming pass nothing,
# it lets an empty-cluster event be settled by the windowed selection (one enqueue, no host round trips)


class QuantizedModel:
    """What the reference reads from the fitted scikit-learn model (utility.py:239):
    ``cluster_centers_`` (K,1) float32, ``labels_`` (N,) int32, ``n_iter_``."""

    def __init__(self, centers: np.ndarray, labels_compact: torch.Tensor, n_iter: int, n_relocations: int,
                 stop: str):
        self.cluster_centers_ = centers.reshape(-1, 1)
        self.labels_compact_ = labels_compact  # device uint8 (K<=256) or int16 storage of uint16
        self.n_iter_ = int(n_iter)
        self.n_relocations_ = int(n_relocations)
        self.stop_reason_ = stop
        self._labels_np = None
        self.counts_device_ = None   # int64[K] index histogram of labels_ (this rank's shard), device
        self.counts_host_ = None     # the same on the host, where the fit's one host read brought it along

    def labels_device(self) -> torch.Tensor:
        """int32 centroid indices on the device."""
        lc = self.labels_compact_
        if lc.dtype == torch.uint8:
            return lc.to(torch.int32)
        return lc.to(torch.int32) & 0xFFFF

    @property
    def labels_(self) -> np.ndarray:
        if self._labels_np is None:
            self._labels_np = self.labels_device().cpu().numpy()
        return self._labels_np


def _allreduce_(t: torch.Tensor, op, group):
    import torch.distributed as dist

    dist.all_reduce(t, op=op, group=group)
    return t


_TLS = threading.local()


def _pinned_landing():
    """One pinned float32[6] + int64[2] per host thread for the statistics read (allocating pinned memory per layer
    costs more than the read)."""
    if getattr(_TLS, "stats", None) is None:
        _TLS.stats = (torch.empty(6, dtype=torch.float32, pin_memory=True), torch.empty(2, dtype=torch.int64, pin_memory=True))
    return _TLS.stats


def _prune_landing():
    """One pinned float32[2] + int64[1] per host thread for what the prune step leaves on the device (sigma, threshold, number
    of zeroed weights): copied out asynchronously right behind the prune kernels, read after the layer's last host read."""
    if getattr(_TLS, "prune", None) is None:
        _TLS.prune = (torch.empty(2, dtype=torch.float32, pin_memory=True), torch.empty(1, dtype=torch.int64, pin_memory=True))
    return _TLS.prune


def _status_landing(nbytes: int) -> torch.Tensor:
    """The pinned status slots, one allocation per host thread (the pinned allocator's bookkeeping per fit costs more
    than a look-in).  Safe to share between the fits of a thread: a look-in is waited for before the next one is
    published, and tickets are unique."""
    buf = getattr(_TLS, "status", None)
    if buf is None or buf.numel() < nbytes:
        buf = _TLS.status = torch.zeros(nbytes, dtype=torch.uint8, pin_memory=True)
    return buf


def _ticket_counter() -> ctypes.c_uint64:
    """This host thread's ticket counter (shared with the library's own look-ins, nnc_kmeans_fit: tickets never repeat)."""
    if getattr(_TLS, "ticket_c", None) is None:
        _TLS.ticket_c = ctypes.c_uint64(0)
    return _TLS.ticket_c


def _next_ticket() -> int:
    c = _ticket_counter()
    c.value += 1
    return int(c.value)


class LayerStats:
    """NumPy-exact mean / variance, min / max (over all weights and over the non-zero ones) and the counts of
    negative and zero weights of one vector (this rank's shard for the counts), fetched with ONE host
    synchronisation.  What the k-means set-up, the sort and the weight distribution need."""

    def __init__(self, x: torch.Tensor, n_total: int | None = None, group=None):
        n = x.numel()
        dev = x.device
        n_total = n if n_total is None else n_total
        if group is None and n > 0:
            # one call enqueues the lot (include/nnc.h, nnc_layer_stats_f32)
            L = nat.load()
            out6 = torch.empty(6, dtype=torch.float32, device=dev)
            signs = torch.empty(2, dtype=torch.int64, device=dev)
            ws_bytes = L.nnc_layer_stats_workspace_bytes(n)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            nat.check(L.nnc_layer_stats_f32(x.data_ptr(), n, out6.data_ptr(), signs.data_ptr(), ws.data_ptr(), ws_bytes, ops._stream(x)))
            pin_f, pin_i = _pinned_landing()
            pin_f.copy_(out6, non_blocking=True)
            pin_i.copy_(signs, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            self.mean, self.var, self.min, self.max, self.min_nonzero, self.max_nonzero = (np.float32(v) for v in pin_f.numpy())
            self.n_negative, self.n_zero = (int(v) for v in pin_i.numpy())
            self.n = n
            return
        mean_d, var_d, _ = ops.moments(x, n_total, group)
        if n > 0:
            mm, signs = ops.minmax_signs(x)
        else:
            mm = torch.tensor([np.inf, -np.inf, np.inf, -np.inf], dtype=torch.float32, device=dev)
            signs = torch.zeros(2, dtype=torch.int64, device=dev)
        if group is not None:
            from . import sharding

            mm = torch.cat([sharding.allreduce_minmax(mm[:2].contiguous(), group), sharding.allreduce_minmax(mm[2:].contiguous(), group)])
        pin_f, pin_i = _pinned_landing()   # (the stream is synchronised before anybody else can use them)
        pin_f.copy_(torch.cat([mean_d, var_d, mm]), non_blocking=True)
        pin_i.copy_(signs, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        self.mean, self.var, self.min, self.max, self.min_nonzero, self.max_nonzero = (np.float32(v) for v in pin_f.numpy())
        self.n_negative, self.n_zero = (int(v) for v in pin_i.numpy())
        self.n = n


def sorted_copy(x: torch.Tensor, stats: LayerStats) -> torch.Tensor:
    """Ascending copy of x (nnc_sort_f32; on a pruned vector only the non-zeros are sorted)."""
    L = nat.load()
    stream = ops._stream(x)
    out = torch.empty_like(x)
    if 4 * stats.n_zero >= x.numel():
        ws_bytes = L.nnc_sort_pruned_workspace_bytes(x.numel(), stats.n_negative, stats.n_zero)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
        nat.check(L.nnc_sort_pruned_f32(x.data_ptr(), x.numel(), stats.n_negative, stats.n_zero, out.data_ptr(),
                                        ws.data_ptr(), ws_bytes, stream))
        return out
    ws_bytes = L.nnc_sort_workspace_bytes(x.numel())
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    nat.check(L.nnc_sort_f32(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), ws_bytes, stream))
    return out


REF_NMAX, REF_KMAX = 4096, 128   # include/nnc.h NNC_REF_NMAX / NNC_REF_KMAX


def reference_fit_applies(n: int, k: int, group=None) -> bool:
    """A short tensor on one GPU: the whole fit runs as one launch in the reference's own arithmetic (fit_reference)."""
    return group is None and k <= n <= REF_NMAX and k <= REF_KMAX


def fit_reference(x: torch.Tensor, init, max_iter: int = MAX_ITER, tol: float = TOL, want_values: bool = True):
    """KMeans(n_clusters=k, init=init[:, None], n_init=1, algorithm="full").fit(x[:, None]) (utility.py:237-238) for a short
    vector, in ONE launch and in scikit-learn's own arithmetic (include/nnc.h, nnc_kmeans_fit_reference_f32): float32
    running sums in sample order, so centres, indices and n_iter_ are the reference's bit for bit.
    Returns (QuantizedModel, values or None) like DeviceKMeans.fit."""
    x = x.reshape(-1)
    ops._require_cuda(x, "x", torch.float32)
    L = nat.load()
    init = np.ascontiguousarray(np.asarray(init, dtype=np.float32).reshape(-1))
    n, k = x.numel(), int(init.size)
    if not reference_fit_applies(n, k):
        raise ValueError(f"fit_reference: n={n}, k={k} outside k <= n <= {REF_NMAX}, k <= {REF_KMAX}")
    dev, stream = x.device, ops._stream(x)
    init_d = ops.small_to_device(init, dev)
    lab = torch.empty(n, dtype=torch.uint8, device=dev)
    vals = torch.empty(n, dtype=torch.float32, device=dev) if want_values else None
    # one device block for everything the host reads back: centres (k float32), result (8 x 4 bytes; the diagnostic build
    # appends 8 int64 phase times)
    kp = (k + 1) & ~1
    blk = torch.empty(8 * k + 4 * (kp + 8 + 16), dtype=torch.uint8, device=dev)   # index histogram (k int64) in front
    counts, back = blk[: 8 * k].view(torch.int64), blk[8 * k:].view(torch.float32)
    nat.check(L.nnc_kmeans_fit_reference_f32(x.data_ptr(), n, init_d.data_ptr(), k, int(max_iter), float(tol), lab.data_ptr(), ops._ptr(vals),
                                             back.data_ptr(), counts.data_ptr(), back.data_ptr() + 4 * kp, stream))
    hostb = blk.cpu().numpy()
    host = hostb[8 * k:].view(np.float32)
    res = host[kp: kp + 8].view(np.int32)
    if not (np.isfinite(host[kp + 6]) and np.isfinite(host[kp + 7])):   # the kernel's own mean / tolerance of the data
        raise ValueError("Input X contains NaN or infinity.")            # (KMeans.fit's input validation)
    model = QuantizedModel(host[:k].copy(), lab, int(res[0]), int(res[2]), {1: "tol", 2: "max_iter", 3: "strict"}.get(int(res[1]), "?"))
    model.counts_device_ = counts
    model.counts_host_ = hostb[: 8 * k].view(np.int64).copy()
    model.n_reloc_windowed_ = 0
    model.reloc_tie_ = int(res[3])
    model.n_reloc_multi_ = int(res[4])
    model.arith_ = "reference"
    model.phase_times_ = host[kp + 8:].view(np.int64)   # (diagnostic build only)
    return model, vals


F32_COUNT_MAX = 1 << 24           # scikit-learn counts members in float32 (weight_in_clusters += 1.0f, _k_means_lloyd.pyx:216): the count of a
#                                   cluster stops at 2^24 (16 777 216 + 1 rounds back to 16 777 216); the pruned zeros of a 25 M tensor get there


def fit_reference_large(x: torch.Tensor, init, max_iter: int = MAX_ITER, tol: float = TOL, want_values: bool = True, stats=None):
    """KMeans(n_clusters=k, init=init[:, None], n_init=1, algorithm="full").fit(x[:, None]) (utility.py:237-238) in scikit-learn's
    own arithmetic for a tensor too long for the one-launch form (fit_reference): opt-in (arith="reference"), slow, exact.

    What makes the reference's centres differ from the exact-sum fit in the last bits is the M-step: float32 running sums in SAMPLE
    order (_k_means_lloyd.pyx:215-218, one thread).  Here they are exactly that (nnc_ref_sums_f32: one wave per cluster walking
    the label vector; an iteration costs the largest cluster times a few nanoseconds -- the zero cluster of a pruned fc1 about a
    millisecond).  The E-step is the device's (bit-exact float32 arg-min, the same as everywhere); everything K-sized --
    float32 member counts (they stop at 2^24, as scikit-learn's do), relocation of empty clusters with ``numpy.argpartition`` itself on the squared distances in sample order
    (_k_means_common.pyx:167-211), averaging by float32(1 / count), the shift, NumPy's pairwise float32 sum, the two stopping
    rules (_kmeans.py:705-733) -- runs in NumPy on the host from 12 bytes per centre per iteration.  One GPU."""
    x = x.reshape(-1)
    ops._require_cuda(x, "x", torch.float32)
    init = np.ascontiguousarray(np.asarray(init, dtype=np.float32).reshape(-1))
    n, k = x.numel(), int(init.size)
    if not (1 <= k <= n):
        raise ValueError(f"fit_reference_large: n={n}, k={k} outside k <= n")
    km = DeviceKMeans(x, init, max_iter=max_iter, tol=tol, sort=False, stats=stats)   # statistics, tolerance, E-step tables; its Lloyd loop is not used
    L, dev, stream = km.L, km.dev, km.stream
    lb = 1 if k <= 256 else 2
    x_mean = np.float32(km.p.x_mean)
    tol_ = np.float32(km.tol_)
    centers = (init - x_mean).astype(np.float32)            # _kmeans.py:1484 (the device has done the same)
    blk = torch.empty(12 * k, dtype=torch.uint8, device=dev)
    counts_d, sums_d = blk[: 8 * k].view(torch.int64), blk[8 * k:].view(torch.float32)   # (the 8-byte words first: alignment)
    cen_d = torch.empty(k, dtype=torch.float32, device=dev)
    labels_old = None
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    n_iter, strict, n_reloc, ties, multi = 0, False, 0, 0, 0
    lab, stop = None, "max_iter"
    for i in range(int(max_iter)):
        lab = km.assign(which=0, labels=True)[0]
        nat.check(L.nnc_ref_sums_f32(x.data_ptr(), n, float(x_mean), lab.data_ptr(), lb, k, sums_d.data_ptr(), counts_d.data_ptr(), stream))
        same = False
        if labels_old is not None:
            nat.check(L.nnc_labels_equal(lab.data_ptr(), labels_old.data_ptr(), n, lb, flag.data_ptr(), stream))
        host = blk.cpu().numpy()
        sums = host[8 * k:].view(np.float32).copy()
        wic = np.minimum(host[: 8 * k].view(np.int64), F32_COUNT_MAX).astype(np.float32)   # scikit-learn's float32 running count, saturation included
        if labels_old is not None:
            same = bool(int(flag.item()))
        empty = np.flatnonzero(wic == 0)
        if empty.size:
            # _relocate_empty_clusters_dense: the reference's own selection, ties and pairing included
            d = km.assign(which=0, labels=False, distances=True)[2].cpu().numpy()
            far = np.argpartition(d, -empty.size)[: -empty.size - 1: -1]
            if np.max(d) != 0:
                n_reloc += 1
                multi += int(empty.size > 1)
                if empty.size < d.size:
                    cut = np.partition(d, [d.size - empty.size - 1, d.size - empty.size])[d.size - empty.size - 1: d.size - empty.size + 1]
                    if cut[0] == cut[1] and cut[1] != 0:   # two samples equally far at the selection cut: introselect decides
                        ties += 1
                idx = torch.from_numpy(far.astype(np.int64)).to(dev)
                xf = (x[idx].cpu().numpy() - x_mean).astype(np.float32)
                lf = lab[idx].cpu().numpy().astype(np.int64) & (0xFF if lb == 1 else 0xFFFF)
                for new, v, old in zip(empty, xf, lf):
                    v = np.float32(v * np.float32(1.0))
                    sums[old] = np.float32(sums[old] - v)
                    sums[new] = v
                    wic[new] = np.float32(1.0)
                    wic[old] = np.float32(wic[old] - np.float32(1.0))
        # _average_centers (_k_means_common.pyx:274-296), in place and in index order
        amax = int(np.argmax(wic))
        cen = sums.copy()
        for j in range(k):
            if wic[j] > 0:
                cen[j] = np.float32(cen[j] * np.float32(1.0 / float(wic[j])))
            else:
                cen[j] = cen[amax]
        # _center_shift (_k_means_common.pyx:298-311) and the total as _kmeans.py:726 has it
        t = (cen - centers).astype(np.float32)
        shift = np.sqrt((t * t).astype(np.float32)).astype(np.float32)
        tot = np.float32((shift ** 2).sum())
        centers = cen
        n_iter = i + 1
        cen_d.copy_(torch.from_numpy(centers))
        nat.check(L.nnc_kmeans_set_centers(km.ws.data_ptr(), ctypes.byref(km.p), cen_d.data_ptr(), 1, stream))
        if same:                       # np.array_equal(labels, labels_old): strict convergence, these labels stay
            strict, stop = True, "strict"
            break
        if tot <= tol_:
            stop = "tol"
            break
        labels_old = lab
    if not strict:
        lab = km.assign(which=0, labels=True)[0]
    final = (centers + x_mean).astype(np.float32)
    vals = None
    if want_values:
        vals = torch.from_numpy(final).to(dev)[lab.to(torch.int64) if lb == 1 else (lab.to(torch.int64) & 0xFFFF)]
    model = QuantizedModel(final, lab, n_iter, n_reloc, stop)
    counts = ops.bincount(lab, k)   # (the library's index histogram, as the default path)
    model.counts_device_ = counts
    model.counts_host_ = counts.cpu().numpy().astype(np.int64)
    model.n_reloc_windowed_ = 0
    model.reloc_tie_ = ties
    model.n_reloc_multi_ = multi
    model.arith_ = "reference"
    return model, vals


def fit_vector(x: torch.Tensor, init, want_values: bool = True, arith: str = "auto", group=None, **kw):
    """The fit behind get_quantized_weight / compress_layer.  ``arith``: "reference" = scikit-learn's float32 running sums
    in sample order, i.e. scikit-learn on ONE OpenMP thread (with more threads its chunk-wise partial sums are reduced in arrival
    order and the result is not reproducible) -- one GPU; one launch for short tensors, fit_reference_large beyond: slow,
    for callers who want the reference's centres bit for bit --, "fixed" = exact fixed-point sums (any size, any number of GPUs), "auto" = reference
    where the one-launch form applies, fixed otherwise."""
    if arith not in ("auto", "reference", "fixed"):
        raise ValueError("arith must be 'auto', 'reference' or 'fixed'")
    k = int(np.asarray(init).size)
    if kw.get("reloc") == "reference" and arith == "auto":
        arith = "fixed"   # (the one-launch fit of a short tensor re-seeds on chip: NumPy's own selection needs the host in the loop)
    if arith == "reference" or (arith == "auto" and reference_fit_applies(x.numel(), k, group)):
        if group is not None:
            raise ValueError("arith='reference' is a single-GPU fit")
        if arith == "reference" and (kw.get("reloc") == "reference" or not reference_fit_applies(x.numel(), k, group)):
            # beyond the one-launch form (or numpy.argpartition's own pairing asked for, which needs the host in the loop): sample-order
            # sums on the device, the K-sized steps in NumPy (slow, exact, opt-in)
            return fit_reference_large(x, init, want_values=want_values, stats=kw.get("stats"), **{a: kw[a] for a in ("max_iter", "tol") if a in kw})
        return fit_reference(x, init, want_values=want_values, **{a: kw[a] for a in ("max_iter", "tol") if a in kw})
    model, vals = DeviceKMeans(x, init, group=group, **kw).fit(want_values=want_values)
    model.arith_ = "fixed"
    return model, vals


class DeviceKMeans:
    """One fit = one instance.  ``x`` float32, 1-D, contiguous, CUDA."""

    def __init__(self, x: torch.Tensor, init, group=None, max_iter: int = MAX_ITER, tol: float = TOL,
                 batch: int = 16, grid_log2: int = 0, replicas_log2: int = -1, sort: bool | None = None,
                 reloc: str = "auto", stats: LayerStats | None = None, x_sorted: torch.Tensor | None = None,
                 n_total: int | None = None, n_min: int | None = None, comm=None, rank_boundaries: bool = True,
                 two_launch: bool = False, loop: bool = False, mass_in_place: bool = False):
        if x.dim() != 1:
            x = x.reshape(-1)
        ops._require_cuda(x, "x", torch.float32)
        self.L = nat.load()
        self.x = x
        self.group = group
        # sharding.RcclComm over the same ranks: the per-iteration exchange then runs inside the C library
        self.comm = comm if group is not None else None
        self.batch = max(1, int(batch))
        self.dev = x.device
        self.stream = ops._stream(x)
        init = np.ascontiguousarray(np.asarray(init, dtype=np.float32).reshape(-1))
        self.k = int(init.size)
        if not (1 <= self.k <= nat.NNC_KMAX - 8):
            raise ValueError(f"number of centroids {self.k} out of range")
        n = x.numel()
        known = n_total is not None and n_min is not None   # the caller has counted the shards already
        if known:
            self.n_min = int(n_min)
        else:
            n_total = n
            self.n_min = n
        if group is not None and not known:
            import torch.distributed as dist

            t = torch.tensor([n, -n], dtype=torch.int64, device=self.dev)
            tm = t.clone()
            _allreduce_(t, dist.ReduceOp.SUM, group)
            _allreduce_(tm, dist.ReduceOp.MAX, group)
            n_total = int(t[0].item())
            self.n_min = -int(tm[1].item())   # the smallest shard: decisions every rank must take alike depend on it
        if n_total < self.k:
            raise ValueError(f"n_samples={n_total} should be >= n_clusters={self.k}.")
        self.n, self.n_total = n, n_total

        # ---- NumPy-exact mean / var, min / max  (one host sync for the whole fit set-up; the caller may have them already)
        if stats is None:
            stats = LayerStats(x, n_total, group)
        self.stats = stats
        mean, var, xmin, xmax = stats.mean, stats.var, stats.min, stats.max
        if not (np.isfinite(mean) and np.isfinite(var) and np.isfinite(xmin) and np.isfinite(xmax)):
            # KMeans.fit's input validation (sklearn/utils/validation.py, reached from utility.py:238): no NaN, no infinity
            raise ValueError("Input X contains NaN or infinity.")
        self.n_negative, self.n_zero = stats.n_negative, stats.n_zero
        self.x_mean = mean
        self.tol_ = np.float32(var * np.float32(tol))  # np.mean(np.var(X, axis=0)) * tol, float32
        lo, hi = np.float32(xmin - mean), np.float32(xmax - mean)  # exact range of the centred data
        absmax = float(max(abs(lo), abs(hi)))
        self.fix_shift = ops.fix_shift(absmax, n_total)

        self.p = nat.KMeansParams(n=n, n_total=n_total, k=self.k, max_iter=int(max_iter), fix_shift=self.fix_shift,
                                  grid_log2=int(grid_log2), replicas_log2=int(replicas_log2), flags=(nat.NNC_KM_TWO_LAUNCH if two_launch else (nat.NNC_KM_LOOP if loop else 0)) | (nat.NNC_KM_MASS_IN_PLACE if mass_in_place else 0),
                                  x_mean=float(mean), tol=float(self.tol_), lo=float(lo), hi=float(hi))
        self.ws_bytes = self.L.nnc_kmeans_workspace_bytes(self.k)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.dev)
        init_d = ops.small_to_device(init, self.dev)
        nat.check(self.L.nnc_kmeans_init(self.ws.data_ptr(), self.ws_bytes, ctypes.byref(self.p), init_d.data_ptr(), self.stream))
        pptr = self.L.nnc_kmeans_partials(self.ws.data_ptr())
        # view of the device partials (2K int64) inside the workspace, for the all-reduce / relocation edits
        off = pptr - self.ws.data_ptr()
        self.partials = self.ws[off: off + 16 * self.k].view(torch.int64)
        # view of the relocation verdict word inside the workspace (all-reduced across ranks in the sharded form)
        foff = self.L.nnc_kmeans_reloc_flag(self.ws.data_ptr()) - self.ws.data_ptr()
        self._reloc_flag = self.ws[foff: foff + 4].view(torch.int32)
        self.n_relocations = 0
        self.n_reloc_windowed = 0   # relocation events settled by the windowed selection
        self.n_reloc_full = 0       # ... by the full distance pass
        self._reloc_scratch = None
        if reloc not in ("auto", "full", "reference"):
            raise ValueError("reloc must be 'auto', 'full' or 'reference'")
        if reloc == "reference" and group is not None:
            raise ValueError("reloc='reference' (NumPy's own selection on the host) is a single-GPU option")
        self.reloc = reloc
        # pinned host landing zones for the small device->host reads (status block, 4096-bin histogram)
        # two slots of (status block, ticket): a look-in alternates between them, so that one may still be in flight
        # (published behind a speculative batch) while the host reads the other
        ssz = ctypes.sizeof(nat.KMeansStatus) + 8
        self._status_pin = _status_landing(2 * ssz)   # shared by the fits of this process: tickets never repeat (_next_ticket)
        self._slot_addr = [self._status_pin.data_ptr() + i * ssz for i in range(2)]
        self._slot_status = [nat.KMeansStatus.from_address(a) for a in self._slot_addr]
        self._slot_ticket = [ctypes.c_uint64.from_address(a + ctypes.sizeof(nat.KMeansStatus)) for a in self._slot_addr]
        self._ticket = 0
        self._hist_pin = None   # pinned landing zone of the full-pass relocation's histogram, made on first use
        # The iterations stream a value-sorted copy (same sums in any order, far fewer LDS atomics);
        # labels, values and relocation distances always come from the original vector.
        if sort is None:
            sort = self.n_min >= SORT_MIN_WEIGHTS
        self.sorted = bool(sort and n > 0)
        self.sorted_everywhere = bool(sort and self.n_min > 0)
        if x_sorted is not None:
            if x_sorted.numel() != n or x_sorted.dtype != torch.float32 or not x_sorted.is_cuda:
                raise ValueError("x_sorted must be a float32 CUDA vector as long as x")
            self.sorted, self.x_iter = True, x_sorted
            self.sorted_everywhere = True   # the caller sorts on every rank or on none
        else:
            self.x_iter = sorted_copy(x, stats) if self.sorted else x
        # On a sorted vector an iteration only looks up the cluster boundaries (include/nnc.h, nnc_kmeans_prefix_build):
        # block prefix sums of the fixed-point images, built once
        self.prefix = None
        if self.sorted and n > 0 and rank_boundaries and (self.x_iter.data_ptr() & 15) == 0:
            self.prefix = torch.empty(int(self.L.nnc_kmeans_prefix_bytes(n)), dtype=torch.uint8, device=self.dev)
            nat.check(self.L.nnc_kmeans_prefix_build(self.x_iter.data_ptr(), ctypes.byref(self.p), self.prefix.data_ptr(), self.stream))
            self.p.prefix_dev = self.prefix.data_ptr()
        # few centres on one GPU: the library runs a whole batch of iterations as ONE launch that stops by itself at
        # convergence or at an empty cluster (include/nnc.h, nnc_kmeans_iterate_publish), so there is nothing to size
        # the whole vector on one GPU, sorted, with prefix sums, up to NNC_KM_LOOP_KMAX centres: the iterations run inside ONE resident
        # workgroup (include/nnc.h, "The Lloyd loop in one workgroup"); two_launch=True keeps the launch-per-iteration forms,
        # loop=True takes the loop whatever K (same results either way)
        self.lloyd = self.prefix is not None and group is None and not two_launch and (loop or self.k <= nat.NNC_KM_LOOP_KMAX)
        self.one_launch = self.prefix is not None and group is None and self.k <= 64 and int(grid_log2) <= 11 and not self.lloyd

    # -------------------------------------------------------------- low-level steps
    def publish(self) -> int:
        """Enqueue a look-in: a one-thread kernel writes the status block into pinned host memory, then a ticket.
        Returns the ticket; wait(ticket) polls for it.  (A few microseconds instead of a copy command plus a stream
        synchronisation; and the host may enqueue more work before it waits.)  At most two may be outstanding."""
        self._ticket = _next_ticket()
        nat.check(self.L.nnc_kmeans_status_publish(self.ws.data_ptr(), self._slot_addr[self._ticket & 1], self._ticket, self.stream))
        return self._ticket

    def wait(self, ticket: int) -> nat.KMeansStatus:
        """Polls the ticket word the device writes behind the status block (pinned, fine-grained coherent host memory:
        torch's pinned allocations are, unless HIP_HOST_COHERENT=0).  A short spin (a look-in normally lands within a
        few hundred microseconds), then the GIL is released between polls so that other host threads run, and after
        20 ms the stream is synchronised once: that makes the write visible even through a non-coherent mapping and
        surfaces a device error instead of hanging."""
        t = self._slot_ticket[ticket & 1]
        spins, synced, t0 = 0, False, 0.0
        while t.value != ticket:
            spins += 1
            if spins == 2000:
                t0 = time.monotonic()
            elif spins > 2000:
                time.sleep(0)   # yield: a second rank thread or a data loader must not starve
                if spins & 0x3F == 0:
                    el = time.monotonic() - t0
                    if el > 0.02 and not synced:
                        torch.cuda.current_stream(self.dev).synchronize()
                        synced = True
                    elif el > 60.0:
                        raise RuntimeError("k-means status never arrived")
        return self._slot_status[ticket & 1]

    def status(self) -> nat.KMeansStatus:
        """The device state after everything enqueued so far."""
        return self.wait(self.publish())

    def loop_stats(self) -> dict:
        """Where the iterations ran so far (include/nnc.h, nnc_kmeans_loop_stats): inside the one-workgroup loop, or handed to the
        multi-workgroup pass.  Synchronises."""
        out = (ctypes.c_int32 * 8)()
        nat.check(self.L.nnc_kmeans_loop_stats(self.ws.data_ptr(), out, self.stream))
        return {"loop_iterations": out[0], "loop_launches": out[1], "reordered": out[2], "handed_over": out[3], "wide_iterations": out[4],
                "relocated_in_loop": out[5], "passed_on_why": out[6], "largest_event": out[7]}

    def iterate_and_look(self, iters: int) -> nat.KMeansStatus:
        """`iters` iterations and the state behind them.  On one GPU the look-in rides on the batch's last launch."""
        if self.group is not None and self.comm is not None:
            self._ticket = _next_ticket()
            nat.check(self.L.nnc_kmeans_iterate_sharded(self.comm.handle, self.x_iter.data_ptr(), self.ws.data_ptr(), ctypes.byref(self.p),
                                                        int(iters), self._slot_addr[self._ticket & 1], self._ticket, self.stream))
            return self.wait(self._ticket)
        if self.group is not None:
            self.iterate(iters)
            return self.status()
        self._ticket = _next_ticket()
        nat.check(self.L.nnc_kmeans_iterate_publish(self.x_iter.data_ptr(), self.ws.data_ptr(), ctypes.byref(self.p), int(iters),
                                                    self._slot_addr[self._ticket & 1], self._ticket, self.stream))
        return self.wait(self._ticket)

    def iterate(self, iters: int):
        """Enqueue `iters` Lloyd iterations (no host sync).  On the resident-loop path (up to NNC_KM_LOOP_KMAX centres) a call asks for
        at most 32 rounds, so more than 32 iterations may need a second call: look at status().iter (include/nnc.h, nnc_kmeans_iterate)."""
        if self.group is None:
            nat.check(self.L.nnc_kmeans_iterate(self.x_iter.data_ptr(), self.ws.data_ptr(), ctypes.byref(self.p), int(iters), self.stream))
            return
        if self.comm is not None:
            nat.check(self.L.nnc_kmeans_iterate_sharded(self.comm.handle, self.x_iter.data_ptr(), self.ws.data_ptr(), ctypes.byref(self.p),
                                                        int(iters), None, 0, self.stream))
            return
        import torch.distributed as dist

        for _ in range(int(iters)):
            nat.check(self.L.nnc_kmeans_accumulate(self.x_iter.data_ptr(), self.ws.data_ptr(), ctypes.byref(self.p), self.stream))
            dist.all_reduce(self.partials, op=dist.ReduceOp.SUM, group=self.group)
            nat.check(self.L.nnc_kmeans_finalize(self.ws.data_ptr(), 0, self.stream))

    def assign(self, which: int = 0, labels: bool = True, values: bool = False, distances: bool = False):
        """E-step on the ORIGINAL vector: labels / cluster_centers_[labels_] / squared distances."""
        return self._assign_on(self.x, which, labels, values, distances)

    def _assign_on(self, src: torch.Tensor, which: int = 0, labels: bool = True, values: bool = False,
                   distances: bool = False, dist_hist: torch.Tensor | None = None):
        lab = q = d = None
        lb = 1 if self.k <= 256 else 2
        if labels:
            lab = torch.empty(self.n, dtype=torch.uint8 if lb == 1 else torch.int16, device=self.dev)
        if values:
            q = torch.empty(self.n, dtype=torch.float32, device=self.dev)
        if distances:
            d = torch.empty(self.n, dtype=torch.float32, device=self.dev)
        nat.check(self.L.nnc_kmeans_assign(src.data_ptr(), self.ws.data_ptr(), ctypes.byref(self.p), int(which),
                                           ops._ptr(lab), lb, ops._ptr(q), ops._ptr(d), ops._ptr(dist_hist), self.stream))
        return lab, q, d

    def centers_device(self, which: int = 0, centred: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(self.k, dtype=torch.float32, device=self.dev)
        nat.check(self.L.nnc_kmeans_get_centers(self.ws.data_ptr(), int(which), 1 if centred else 0, out.data_ptr(), self.stream))
        return out

    def centers(self, which: int = 0, centred: bool = False) -> np.ndarray:
        return self.centers_device(which, centred).cpu().numpy()

    # -------------------------------------------------------------- empty-cluster relocation
    TOPM_CAP = 1 << 16

    def _top_keys(self, d: torch.Tensor, x: torch.Tensor, m: int, hist0: torch.Tensor | None = None) -> torch.Tensor:
        """Keys of the m + 1 samples farthest from their own centre (this shard), descending (the runner-up lets the
        relocation kernel see a tie at the cut; fewer if the shard is shorter).  A key is
        (float32 bits of d) << 32 | order-preserving bits of x: descending keys = descending
        distance, equal distances by descending value; samples equal in both are interchangeable,
        so the outcome does not depend on sample order or on how the vector is sharded.
        Histogram of the distance bits -> threshold -> compaction of the survivors (HIP kernels,
        refined inside the cut bin while it is crowded) -> sort of the few survivors."""
        n = d.numel()
        m = min(m + 1, n)
        if m == 0:
            return torch.empty(0, dtype=torch.int64, device=self.dev)
        hist = hist0 if hist0 is not None else torch.empty(4096, dtype=torch.int64, device=self.dev)
        prefix, decided, thr_bits, cand = None, 0, 0, n
        for shift, width, pshift in ((19, 12, -1), (7, 12, 19), (0, 7, 7)):   # 12 + 12 + 7 value bits (sign bit is 0)
            if not (shift == 19 and hist0 is not None):  # the first level may come for free from the distance pass
                nat.check(self.L.nnc_topm_hist_f32(d.data_ptr(), n, shift, width, pshift, 0 if prefix is None else prefix,
                                                   hist.data_ptr(), self.stream))
            if self._hist_pin is None:
                self._hist_pin = torch.empty(4096, dtype=torch.int64, pin_memory=True)
            self._hist_pin.copy_(hist, non_blocking=True)
            torch.cuda.current_stream(self.dev).synchronize()
            h = self._hist_pin.numpy()
            above = np.cumsum(h[::-1])[::-1] + decided           # samples at or above bin b (within the prefix) + those above the prefix
            ok = np.nonzero(above >= m)[0]
            b = int(ok[-1]) if ok.size else 0                    # highest bin that still leaves >= m samples
            cand = int(above[b])
            decided = int(above[b + 1]) if b + 1 < above.size else decided
            prefix = b if prefix is None else ((prefix << width) | b)
            thr_bits = prefix << shift
            if cand <= self.TOPM_CAP:
                break
        if cand <= self.TOPM_CAP:
            keys = torch.empty(cand, dtype=torch.int64, device=self.dev)
            cnt = torch.zeros(1, dtype=torch.int64, device=self.dev)
            nat.check(self.L.nnc_topm_compact_f32(d.data_ptr(), x.data_ptr(), n, thr_bits, keys.data_ptr(), cand, cnt.data_ptr(), self.stream))
            return torch.sort(keys, descending=True).values[:m]
        # a crowd of exactly equal distances at the cut: general selection over all keys
        xb = x.view(torch.int32).to(torch.int64)
        ordered = torch.where(xb < 0, (~xb) & 0xFFFFFFFF, xb | 0x80000000)
        key = (d.view(torch.int32).to(torch.int64) << 32) | ordered
        return torch.topk(key, m, largest=True, sorted=True).values

    def _relocate_windowed(self, n_empty: int) -> bool:
        """The same relocation from windows of candidates around the cluster boundaries of the
        value-sorted vector (include/nnc.h, nnc_kmeans_relocate_windowed): no pass over the
        vector, no host read.  The device proves the selection; if it cannot, the resumed finalize
        leaves status.paused = 2 and the caller comes back through the full pass.
        Sharded: every rank selects from its own shard, the ranks exchange keys and verdicts."""
        window = int(self.L.nnc_kmeans_reloc_window(self.n_min, n_empty))   # the same decision on every rank
        if window == 0:
            return False
        need = int(self.L.nnc_kmeans_reloc_scratch_bytes(self.k, window))
        if self._reloc_scratch is None or self._reloc_scratch.numel() < need:
            self._reloc_scratch = torch.empty(need, dtype=torch.uint8, device=self.dev)
        ws = self.ws.data_ptr()
        if self.group is None:
            nat.check(self.L.nnc_kmeans_relocate_windowed(self.x_iter.data_ptr(), ws, ctypes.byref(self.p), n_empty,
                                                          self._reloc_scratch.data_ptr(), self._reloc_scratch.numel(), self.stream))
            return True
        if self.comm is not None:
            need = int(self.L.nnc_kmeans_reloc_scratch_bytes_sharded(self.k, window, self.comm.world))
            if self._reloc_scratch.numel() < need:
                self._reloc_scratch = torch.empty(need, dtype=torch.uint8, device=self.dev)
            nat.check(self.L.nnc_kmeans_relocate_windowed_sharded(self.comm.handle, self.x_iter.data_ptr(), ws, ctypes.byref(self.p), n_empty,
                                                                  self._reloc_scratch.data_ptr(), self._reloc_scratch.numel(), self.stream))
            return True
        import torch.distributed as dist

        keys = torch.empty(n_empty + 1, dtype=torch.int64, device=self.dev)
        nat.check(self.L.nnc_kmeans_reloc_select_local(self.x_iter.data_ptr(), ws, ctypes.byref(self.p), n_empty,
                                                       self._reloc_scratch.data_ptr(), self._reloc_scratch.numel(),
                                                       keys.data_ptr(), self.stream))
        dist.all_reduce(self._reloc_flag, op=dist.ReduceOp.MAX, group=self.group)   # any rank unproven -> nobody relocates
        bufs = [torch.empty_like(keys) for _ in range(dist.get_world_size(self.group))]
        dist.all_gather(bufs, keys, group=self.group)
        # (every rank's list is descending: the library's own merge, the one the RCCL path runs inside nnc_kmeans_relocate_windowed_sharded)
        merged = torch.empty(n_empty + 1, dtype=torch.int64, device=self.dev)
        nat.check(self.L.nnc_merge_keys(torch.cat(bufs).data_ptr(), len(bufs), n_empty + 1, merged.data_ptr(), n_empty + 1, self.stream))
        nat.check(self.L.nnc_kmeans_relocate_if_proven(ws, merged.data_ptr(), n_empty + 1, self.stream))
        nat.check(self.L.nnc_kmeans_finalize(ws, 1, self.stream))
        return True

    def _relocate_and_resume(self, st) -> None:
        """scikit-learn's _relocate_empty_clusters_dense (_k_means_common.pyx:167-211) for a
        paused iteration, then resume the finalize step; everything runs on the device (one small
        host read, the 4096-bin distance histogram).  If the labels of this iteration equal those
        of the previous one the state is marked done = 3 (strict convergence).

        The i-th empty cluster (ascending index) receives the i-th farthest sample in the order
        "descending distance, equal distances by descending value".  scikit-learn pairs them in the
        order numpy.argpartition happens to leave the top of its index array in: the same set of
        samples (up to ties at the cut), the same pairing whenever n_empty == 1, an implementation-
        defined (CPU-dispatch dependent) pairing otherwise."""
        n_empty = int(st.n_empty)
        strict_check = st.iter >= 1 and st.same_counts
        if self.reloc == "reference":
            self._relocate_like_numpy(st, n_empty, strict_check)
            return
        if (self.reloc == "auto" and self.sorted_everywhere and int(st.paused) == 1 and not strict_check
                and self._relocate_windowed(n_empty)):
            self.n_relocations += 1
            self.n_reloc_windowed += 1   # provisional: a failed proof comes back as paused == 2 and is redone in full
            return
        if int(st.paused) == 2:
            self.n_relocations -= 1
            self.n_reloc_windowed -= 1
        self.n_reloc_full += 1
        xs = self.x_iter  # any order will do; the value-sorted copy makes the histogram cheap
        hist0 = torch.empty(4096, dtype=torch.int64, device=self.dev)
        _, _, d = self._assign_on(xs, which=0, labels=False, distances=True, dist_hist=hist0)
        flag = None
        if st.iter >= 1 and st.same_counts:
            # labels can only equal the previous iteration's if no cluster changed size
            lab, _, _ = self._assign_on(xs, which=0, labels=True)
            prev, _, _ = self._assign_on(xs, which=1, labels=True)
            lb = 1 if self.k <= 256 else 2
            flag = torch.empty(1, dtype=torch.int32, device=self.dev)
            nat.check(self.L.nnc_labels_equal(lab.data_ptr(), prev.data_ptr(), self.n, lb, flag.data_ptr(), self.stream))
            if self.group is not None:
                import torch.distributed as dist

                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        keys = self._top_keys(d, xs, n_empty, hist0)
        if self.group is not None:
            import torch.distributed as dist

            world = dist.get_world_size(self.group)
            pad = torch.full((n_empty + 1,), -1, dtype=torch.int64, device=self.dev)
            pad[: keys.numel()] = keys
            bufs = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(bufs, pad, group=self.group)
            m_all = min(n_empty + 1, self.n_total)
            keys = torch.empty(m_all, dtype=torch.int64, device=self.dev)   # (descending lists padded with -1: the library's merge; 0 where there are fewer keys)
            nat.check(self.L.nnc_merge_keys(torch.cat(bufs).data_ptr(), world, n_empty + 1, keys.data_ptr(), m_all, self.stream))
        nat.check(self.L.nnc_kmeans_relocate(self.ws.data_ptr(), keys.data_ptr(), int(keys.numel()), self.stream))
        nat.check(self.L.nnc_kmeans_finalize(self.ws.data_ptr(), 1, self.stream))
        if flag is not None:
            nat.check(self.L.nnc_kmeans_set_done_if(self.ws.data_ptr(), flag.data_ptr(), 3, self.stream))
        self.n_relocations += 1

    def _relocate_like_numpy(self, st, n_empty: int, strict_check: bool) -> None:
        """The event exactly as scikit-learn runs it (_k_means_common.pyx:167-211): the float32 squared distances of ALL samples to
        their own centre, in SAMPLE order, go to the host and ``numpy.argpartition(distances, -n_empty)[:-n_empty-1:-1]`` picks the far
        samples -- which ones at a tie at the cut, and which of them goes to which empty cluster, is whatever NumPy's introselect
        leaves, and the reference inherits exactly that.  The device's own rule (descending distance, ties by descending value)
        agrees on 69 of the 70 golden fits; this option is for the callers who want the 70th as well: one read of 4 bytes per
        weight per event (events are rare), everything else stays on the device."""
        lab = prev = None
        _, _, d = self._assign_on(self.x, which=0, labels=False, distances=True)
        flag = None
        if strict_check:
            lab, _, _ = self._assign_on(self.x, which=0, labels=True)
            prev, _, _ = self._assign_on(self.x, which=1, labels=True)
            flag = torch.empty(1, dtype=torch.int32, device=self.dev)
            nat.check(self.L.nnc_labels_equal(lab.data_ptr(), prev.data_ptr(), self.n, 1 if self.k <= 256 else 2, flag.data_ptr(), self.stream))
        dh = d.cpu().numpy()
        far = np.argpartition(dh, -n_empty)[:-n_empty - 1:-1]
        xh = self.x[torch.from_numpy(far.astype(np.int64)).to(self.dev)].cpu().numpy()
        db = dh[far].view(np.uint32).astype(np.int64)
        if float(dh.max()) == 0.0:
            db[:] = 0                                   # np.max(distances) == 0: nothing is relocated (the kernel reads it off the first key)
        xb = xh.view(np.uint32).astype(np.int64)
        ordered = np.where(xb & 0x80000000, (~xb) & 0xFFFFFFFF, xb | 0x80000000)
        keys = torch.from_numpy(((db << 32) | ordered).astype(np.int64)).to(self.dev)
        nat.check(self.L.nnc_kmeans_relocate(self.ws.data_ptr(), keys.data_ptr(), int(keys.numel()), self.stream))
        nat.check(self.L.nnc_kmeans_finalize(self.ws.data_ptr(), 1, self.stream))
        if flag is not None:
            nat.check(self.L.nnc_kmeans_set_done_if(self.ws.data_ptr(), flag.data_ptr(), 3, self.stream))
        self.n_relocations += 1
        self.n_reloc_full += 1

    # -------------------------------------------------------------- the fit loop
    def fit(self, want_values: bool = True):
        """Runs to convergence.  Returns (QuantizedModel, values tensor or None) where
        values = cluster_centers_[labels_] as a device float32 vector (utility.py:239)."""
        strict_labels = None
        if self.group is None:
            # one GPU: the loop itself -- batches, look-ins, windowed relocations -- runs inside the library (nnc_kmeans_fit); it
            # comes back when the fit has stopped or an empty-cluster event needs the full-pass relocation
            st = nat.KMeansStatus()
            nwin = ctypes.c_int32(0)
            if self.sorted_everywhere and self.reloc == "auto" and not self.one_launch and self.prefix is not None and self.n >= 512:
                # room for the relocation chain the library enqueues behind the first iterations in case they pause for
                # empty clusters (windows of up to 256 samples: include/nnc.h, nnc_kmeans_fit)
                need = int(self.L.nnc_kmeans_reloc_scratch_bytes(self.k, 256))
                if self._reloc_scratch is None or self._reloc_scratch.numel() < need:
                    self._reloc_scratch = torch.empty(need, dtype=torch.uint8, device=self.dev)
            while True:
                scratch = self._reloc_scratch
                nat.check(self.L.nnc_kmeans_fit(self.x_iter.data_ptr(), self.ws.data_ptr(), ctypes.byref(self.p), self.batch,
                                                1 if (self.sorted_everywhere and self.reloc == "auto") else 0,
                                                ops._ptr(scratch), 0 if scratch is None else scratch.numel(), self._status_pin.data_ptr(),
                                                ctypes.byref(_ticket_counter()), ctypes.byref(st), ctypes.byref(nwin), self.stream))
                self.n_relocations += nwin.value
                self.n_reloc_windowed += nwin.value
                if st.done:
                    break
                self._relocate_and_resume(st)
        if self.group is not None and self.comm is not None:
            # sharded over the library's own communicator: the same, one call per rank (nnc_kmeans_fit_sharded)
            st = nat.KMeansStatus()
            nwin = ctypes.c_int32(0)
            windowed = self.sorted_everywhere and self.reloc == "auto"
            if windowed and self.n_min >= 512:
                # (windows of up to 256 samples; a larger event comes back to _relocate_windowed, which grows the block for good)
                need = int(self.L.nnc_kmeans_reloc_scratch_bytes_sharded(self.k, 256, self.comm.world))
                if self._reloc_scratch is None or self._reloc_scratch.numel() < need:
                    self._reloc_scratch = torch.empty(need, dtype=torch.uint8, device=self.dev)
            while True:
                scratch = self._reloc_scratch
                nat.check(self.L.nnc_kmeans_fit_sharded(self.comm.handle, self.x_iter.data_ptr(), self.ws.data_ptr(), ctypes.byref(self.p),
                                                        self.n_min, self.batch, 1 if windowed else 0,
                                                        ops._ptr(scratch), 0 if scratch is None else scratch.numel(), self._status_pin.data_ptr(),
                                                        ctypes.byref(_ticket_counter()), ctypes.byref(st), ctypes.byref(nwin), self.stream))
                self.n_relocations += nwin.value
                self.n_reloc_windowed += nwin.value
                if st.done:
                    break
                self._relocate_and_resume(st)
        batch = MAX_ITER if self.one_launch else 1  # the first iteration is where duplicate initial centres surface as empty clusters
        hist = []  # (iteration, sum of squared centre shifts) at the host's look-ins
        while self.group is not None and self.comm is None:
            st = self.iterate_and_look(batch)
            if st.done:
                break
            if st.paused:
                # an empty cluster stopped the device loop inside this batch: relocate and resume that
                # iteration on the device, then go on one iteration at a time for a while.  No look-in
                # in between: if the resumed iteration was the last one the next launch is a no-op.
                self._relocate_and_resume(st)
                batch = MAX_ITER if self.one_launch else 1
                hist = []
                continue
            # size the next batch so that it ends about where the shift crosses the tolerance
            # (launches enqueued after convergence are no-ops, but they still cost a dispatch)
            hist.append((int(st.iter), float(st.shift_tot)))
            if self.one_launch:
                continue
            batch = min(self.batch, batch * 2)
            if len(hist) >= 2 and hist[-1][1] > 0 and hist[-2][1] > hist[-1][1] and self.tol_ > 0:
                (i0, s0), (i1, s1) = hist[-2], hist[-1]
                rate = math.log(s0 / s1) / max(1, i1 - i0)          # log-decay per iteration
                left = math.log(s1 / float(self.tol_)) / rate if s1 > float(self.tol_) else 0.0
                batch = int(max(1, min(self.batch, math.floor(left * 0.9))))
        if int(st.done) == 3:
            # strict stop: keep the labels of that iteration = E-step on the centres it started
            # from, which the resumed finalize has made the "previous" set
            strict_labels = self.assign(which=1, labels=True)[0]
        stop = {1: "tol", 2: "max_iter", 3: "strict"}.get(int(st.done), "?")
        n_iter = int(st.iter)
        # one device block for what the host reads at the end: index histogram (k int64), centres (k float32)
        blk = torch.empty(12 * self.k, dtype=torch.uint8, device=self.dev)
        counts, cen_d = blk[: 8 * self.k].view(torch.int64), blk[8 * self.k:].view(torch.float32)
        self.centers_device(which=0, centred=False, out=cen_d)
        if strict_labels is not None:
            # label-equality stop: scikit-learn keeps the labels of that iteration and does
            # not run another E-step (_kmeans.py:717-722, 736)
            lab = strict_labels
            vals = None
            if want_values:
                idx = lab.to(torch.int64) if lab.dtype == torch.uint8 else (lab.to(torch.int64) & 0xFFFF)
                vals = cen_d[idx]
        else:
            lab, vals, _ = self.assign(which=0, labels=True, values=want_values)
        # index histogram of these labels (this rank's shard), from one more pass over the iteration copy
        nat.check(self.L.nnc_kmeans_label_counts(self.x_iter.data_ptr(), self.ws.data_ptr(), ctypes.byref(self.p),
                                                 1 if strict_labels is not None else 0, counts.data_ptr(), self.stream))
        host = blk.cpu().numpy()   # the one host read of the epilogue, behind everything that was enqueued
        model = QuantizedModel(host[8 * self.k:].view(np.float32).copy(), lab, n_iter, self.n_relocations, stop)
        model.counts_device_ = counts
        model.counts_host_ = host[: 8 * self.k].view(np.int64).copy()   # (this rank's shard of the vector)
        model.n_reloc_windowed_ = self.n_reloc_windowed   # relocation events settled without a pass over the vector
        # what scikit-learn leaves to numpy.argpartition (include/nnc.h, nnc_kmeans_status): events in which two different
        # samples tied at the selection cut (the fits may part ways there), events with more than one empty cluster
        model.reloc_tie_ = int(st.reloc_ties)
        model.n_reloc_multi_ = int(st.reloc_multi)
        model.arith_ = "fixed"
        return model, vals
