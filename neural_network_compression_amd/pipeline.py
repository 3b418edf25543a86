"""The per-layer Deep-Compression step on data resident in HBM: prune -> (CDF) -> k-means
-> index histogram -> Huffman code lengths, single GPU or one shard per GPU.

This is what the reference does to one layer tensor across ``Trainer._prune_parameters``
(common/trainer.py:177-193) and ``Trainer.quantize`` (common/trainer.py:42-72), with the
Huffman length histogram the north star adds (the reference stops at the index gather).
"""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass

import numpy as np
import torch

from . import kmeans as _kmeans
from . import ops, sharding
from .common import utility


@dataclass
class LayerResult:
    mask: torch.Tensor | None      # uint8/bool mask of pruned entries (this shard), or None if not pruned
    nzeroed: int | None
    sigma: float | None
    threshold: float | None
    values: torch.Tensor | None    # quantized weights (this shard), float32
    model: object | None           # kmeans.QuantizedModel, None if the tensor passed through
    counts: np.ndarray | None      # int64[K] index histogram over the whole vector
    code_lengths: np.ndarray | None
    length_hist: np.ndarray | None
    total_bits: int | None


def prune_sharded_(x: torch.Tensor, q, std_smooth: bool, group=None, n_total: int | None = None):
    """prune_weigth on this rank's shard of a longer vector.  Returns (mask, stats, nzeroed)."""
    if group is None:
        return ops.prune_(x, q, std_smooth)
    n_total = sharding.total_count(x.numel(), x.device, group) if n_total is None else n_total
    if std_smooth:
        _, _, std = ops.moments(x, n_total, group)
        thr = std * float(np.float32(q))  # float32 * float32 on the device, as np.std(w) * q
    else:
        std = torch.zeros(1, dtype=torch.float32, device=x.device)
        thr = torch.tensor([np.float32(q)], dtype=torch.float32, device=x.device)
    mask, nz = ops.threshold_mask_(x, thr)
    sharding.allreduce_sum_(nz, group)
    return mask, torch.cat([std, thr]), nz


def weight_distribution(x: torch.Tensor, skip_zeros: bool = True, group=None):
    """get_weight_distribution over the whole (possibly sharded) vector."""
    if group is None:
        return utility._weight_distribution_device(x, skip_zeros)
    mm, cnt = ops.minmax(x, skip_zeros=skip_zeros)
    mm = sharding.allreduce_minmax(mm, group)
    sharding.allreduce_sum_(cnt, group)
    host = mm.cpu().numpy()
    if int(cnt.item()) == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    steps = np.linspace(np.float32(host[0]), np.float32(host[1]), num=32)
    steps_d = torch.from_numpy(np.ascontiguousarray(steps, dtype=np.float32)).to(x.device)
    counts = sharding.allreduce_sum_(ops.hist31(x, steps_d, skip_zeros=skip_zeros), group).cpu().numpy()
    return utility._cdf_from_counts(steps, counts)


def weight_distribution_sorted(x_sorted: torch.Tensor, stats, group=None):
    """get_weight_distribution of the non-zero weights (utility.py:334-392, Trainer.quantize strips the zeros
    first) from a value-sorted copy: the 31 bin counts are differences of 32 ranks (one binary search each)
    instead of a pass over the vector.  Same float32 comparisons as the histogram kernel: bin b holds
    steps[b] <= w < steps[b+1].  Sharded: every rank ranks the steps in its own sorted shard, the ranks add up."""
    if not np.isfinite(stats.min_nonzero):   # no non-zero weight anywhere (min over nothing = +inf)
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    steps = np.linspace(np.float32(stats.min_nonzero), np.float32(stats.max_nonzero), num=32)  # float32 under NumPy 2
    steps32 = np.ascontiguousarray(steps, dtype=np.float32)
    steps_d = ops.small_to_device(steps32, x_sorted.device)
    from . import _native as nat

    t = torch.empty(33, dtype=torch.int64, device=x_sorted.device)   # 32 ranks: #{w < steps[b]} in this shard; then this shard's zeros
    nat.check(nat.load().nnc_rank_sorted_f32(x_sorted.data_ptr(), x_sorted.numel(), steps_d.data_ptr(), 32, t.data_ptr(), ops._stream(x_sorted)))
    if group is not None:
        t[32] = int(stats.n_zero)
        sharding.allreduce_sum_(t, group)
    host = t.cpu().numpy()
    if group is None:
        host[32] = int(stats.n_zero)
    counts = np.diff(host[:32])
    zero = np.float32(0.0)
    inside = np.nonzero((steps32[:-1] <= zero) & (zero < steps32[1:]))[0]
    if inside.size:
        counts[inside[0]] -= int(host[32])   # the zeros sit in that bin of the full vector
    return utility._cdf_from_counts(steps, counts)


def initial_centroids(x: torch.Tensor, bits: int, mode: str, cdfs=None, group=None, n_total=None) -> np.ndarray:
    """The reference's init `space` (utility.py:206-226) for a possibly sharded vector."""
    if group is None:
        return np.asarray(utility._init_space(x, x.numel(), bits, mode, cdfs), dtype=np.float32)
    if mode == "linear":
        mm, _ = ops.minmax(x)
        host = sharding.allreduce_minmax(mm, group).cpu().numpy()
        return np.linspace(np.float32(host[0]), np.float32(host[1]), num=2 ** bits).astype(np.float32)
    if mode == "density" and cdfs is not None:
        return np.asarray(utility._init_space(x, x.numel(), bits, mode, cdfs), dtype=np.float32)
    if mode == "forgy":
        import torch.distributed as dist

        # every rank draws (so that every rank's global RNG advances as the reference's would), but only rank 0's draw
        # counts: the ranks' RNG states are not guaranteed to agree; each rank contributes the samples it owns
        n_total = sharding.total_count(x.numel(), x.device, group) if n_total is None else n_total
        idx = np.random.randint(0, n_total, size=2 ** bits)
        idx_t = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(x.device)
        dist.broadcast(idx_t, src=dist.get_global_rank(group, 0), group=group)
        idx = idx_t.cpu().numpy()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        lo, hi = sharding.shard_bounds(n_total, world, rank)
        vals = torch.zeros(idx.size, dtype=torch.float32, device=x.device)
        mine = np.nonzero((idx >= lo) & (idx < hi))[0]
        if mine.size:
            vals[torch.from_numpy(mine).to(x.device)] = x[torch.from_numpy(idx[mine] - lo).to(x.device)]
        sharding.allreduce_sum_(vals, group)  # exactly one rank contributes each entry
        return vals.cpu().numpy()
    raise Exception(" error mode not found")


# ------------------------------------------------------------------ the whole layer as one call into the library
_LAYER_TLS = threading.local()


def _layer_host_block():
    """This host thread's pinned block for nnc_compress_layer_f32 (status slots, scalars, the K-sized read) and its ticket
    counter, which lives as long as the block."""
    blk = getattr(_LAYER_TLS, "blk", None)
    if blk is None:
        import ctypes

        from . import _native as nat

        nb = int(nat.load().nnc_compress_layer_host_bytes())
        blk = _LAYER_TLS.blk = (torch.zeros(nb, dtype=torch.uint8, pin_memory=True), ctypes.c_uint64(0), nat.LayerResult())
    return blk


def _compress_layer_native(x: torch.Tensor, q, std_smooth: bool, bits: int, mode: str, want_values: bool, two_launch: bool = False, loop: bool = False):
    """nnc_compress_layer_f32 (include/nnc.h) on a whole tensor of one GPU, modes linear / density.  Returns a LayerResult, or
    (mask, sigma, threshold, nzeroed) of the pruned tensor when the library leaves the fit to the step-by-step path."""
    import ctypes

    from . import _native as nat

    L = nat.load()
    n = x.numel()
    k = 2 ** bits + (1 if mode == "density" else 0)
    dev = x.device
    ws_bytes = int(L.nnc_compress_layer_workspace_bytes(n, k))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev) if q is not None else None
    labels = torch.empty(n, dtype=torch.uint8 if k <= 256 else torch.int16, device=dev)
    values = torch.empty(n, dtype=torch.float32, device=dev) if want_values else None
    pinned, ticket, res = _layer_host_block()
    lp = nat.LayerParams(q=float(np.float32(q)) if q is not None else 0.0, prune=1 if q is not None else 0, std_smooth=1 if std_smooth else 0,
                         bits=int(bits), mode=1 if mode == "density" else 0, want_values=1 if want_values else 0,
                         km_flags=nat.NNC_KM_TWO_LAUNCH if two_launch else (nat.NNC_KM_LOOP if loop else 0))
    nat.check(L.nnc_compress_layer_f32(x.data_ptr(), n, ctypes.byref(lp), ops._ptr(mask), labels.data_ptr(), ops._ptr(values), ws.data_ptr(), ws_bytes,
                                       pinned.data_ptr(), pinned.numel(), ctypes.byref(ticket), ctypes.byref(res), ops._stream(x)))
    mask_b = mask   # (uint8, like ops.prune_)
    sigma = thr = nz = None
    if q is not None:
        sigma, thr, nz = float(res.sigma), float(res.threshold), int(res.n_zeroed)
    if res.status != 0:
        return mask_b, sigma, thr, nz
    centers = np.frombuffer(res.centers, dtype=np.float32, count=k).copy()
    counts = np.frombuffer(res.counts, dtype=np.int64, count=k).copy()
    lengths = np.frombuffer(res.code_lengths, dtype=np.uint8, count=k).copy()
    model = _kmeans.QuantizedModel(centers, labels, int(res.n_iter), int(res.n_relocations), {1: "tol", 2: "max_iter", 3: "strict"}.get(int(res.stop), "?"))
    model.counts_host_ = counts
    model.n_reloc_windowed_ = int(res.n_reloc_windowed)
    model.reloc_tie_ = int(res.reloc_ties)
    model.n_reloc_multi_ = int(res.reloc_multi)
    model.arith_ = "reference" if res.arith == 1 else "fixed"
    top = int(lengths.max()) if k else 0
    lhist = np.bincount(lengths, minlength=top + 1).astype(np.int64)
    return LayerResult(mask_b, nz, sigma, thr, values, model, counts, lengths, lhist, int(res.total_bits))


def compress_layer(x: torch.Tensor, q=None, std_smooth: bool = True, bits: int = 4, mode: str = "linear",
                   with_cdf: bool | None = None, group=None, huffman: bool = True,
                   want_values: bool = True, comm=None, arith: str = "auto", native: bool = True,
                   two_launch: bool = False, loop: bool = False) -> LayerResult:
    """One layer tensor (or this rank's shard of it), in place on `x` for the pruning part.
    ``group``: torch.distributed group of one rank per GPU when `x` is a shard; ``comm`` (sharding.RcclComm over the same
    ranks) moves the per-iteration exchange of the fit into the C library.  ``arith``: kmeans.fit_vector.  ``native``: let
    the library run the whole layer as one call where it can (same results; ``False``: the step-by-step path).
    ``two_launch``: iterate launch by launch instead of inside one resident workgroup (include/nnc.h, NNC_KM_TWO_LAUNCH; same results);
    ``loop``: the resident workgroup whatever the number of centres (NNC_KM_LOOP; the library's own choice otherwise)."""
    x = x.reshape(-1)
    ops._require_cuda(x, "x", torch.float32)
    n_total = n_min = x.numel()
    if group is not None:
        import torch.distributed as dist

        sizes = [torch.zeros(1, dtype=torch.int64, device=x.device) for _ in range(dist.get_world_size(group))]
        dist.all_gather(sizes, torch.tensor([x.numel()], dtype=torch.int64, device=x.device), group=group)
        sizes = torch.cat(sizes).cpu().numpy()
        n_total, n_min = int(sizes.sum()), int(sizes.min())
    mask = nz = sigma = thr = None
    early = None
    pre = None
    if (native and group is None and arith == "auto" and mode in ("linear", "density") and huffman and with_cdf in (None, mode == "density")
            and 1 <= bits <= 10 and n_total >= (2 ** bits) + 1 and x.is_contiguous()):
        # the whole layer as one call into the library (nnc_compress_layer_f32); it hands the fit back when that needs the
        # step-by-step path (short tensor with the density init, full-pass relocation, strict-convergence check)
        out = _compress_layer_native(x, q, std_smooth, bits, mode, want_values, two_launch, loop)
        if isinstance(out, LayerResult):
            return out
        pre = out          # (mask, sigma, threshold, nzeroed): the tensor is pruned already
        q = None
    if q is not None:
        mask, stats, nzt = prune_sharded_(x, q, std_smooth, group, n_total)
        if group is None:
            early = _kmeans._prune_landing()
            early[0].copy_(stats.reshape(-1)[:2], non_blocking=True)
            early[1].copy_(nzt.reshape(-1)[:1], non_blocking=True)
    if n_total < (2 ** bits) + 1:
        print("not enough bits:", n_total, " vs ", 2 ** bits)
        if q is not None:
            s = stats.cpu().numpy()
            sigma, thr, nz = float(s[0]), float(s[1]), int(nzt.item())
        return LayerResult(mask, nz, sigma, thr, x, None, None, None, None, None)
    if with_cdf is None:
        with_cdf = mode == "density"
    lstats = x_sorted = None
    k = 2 ** bits + (1 if mode == "density" else 0)
    short = arith == "reference" or (arith == "auto" and _kmeans.reference_fit_applies(n_total, k, group))
    if n_min >= _kmeans.SORT_MIN_WEIGHTS and not short:
        # long vector (every shard of it): one statistics pass and one sort serve the weight distribution, the init and the fit
        lstats = _kmeans.LayerStats(x, n_total, group)
        x_sorted = _kmeans.sorted_copy(x, lstats)
        cdfs = weight_distribution_sorted(x_sorted, lstats, group) if with_cdf else None
    else:
        cdfs = weight_distribution(x, skip_zeros=True, group=group) if with_cdf else None
    if mode == "linear" and lstats is not None:
        space = np.linspace(np.float32(lstats.min), np.float32(lstats.max), num=2 ** bits).astype(np.float32)
    else:
        space = initial_centroids(x, bits, mode, cdfs, group, n_total)
    if short:
        model, values = _kmeans.fit_reference(x, space, want_values=want_values)
    else:
        km = _kmeans.DeviceKMeans(x, space, group=group, stats=lstats, x_sorted=x_sorted, n_total=n_total, n_min=n_min, comm=comm,
                                  two_launch=two_launch, loop=loop)
        model, values = km.fit(want_values=want_values)
    k = int(model.cluster_centers_.size)
    counts = lengths = lhist = total = None
    if huffman:
        counts = getattr(model, "counts_host_", None) if group is None else None   # came along with the fit's host read
        if counts is None:
            counts_d = getattr(model, "counts_device_", None)
            if counts_d is None:
                counts_d = ops.bincount(model.labels_compact_, k)
            if group is not None:
                sharding.allreduce_sum_(counts_d, group)
            counts = counts_d.cpu().numpy()
        lengths, lhist, total = ops.huffman_lengths(counts)
    if q is not None:
        if early is not None:
            # sigma / threshold / number of zeroed weights left the device right behind the prune kernels; the fit's host read
            # (a synchronisation of this stream) lies behind them
            sigma, thr, nz = float(early[0][0]), float(early[0][1]), int(early[1][0])
        else:
            sf = stats.reshape(-1)[:2].cpu().numpy()
            sigma, thr, nz = float(sf[0]), float(sf[1]), int(nzt.item())
    if pre is not None:
        mask, sigma, thr, nz = pre
    return LayerResult(mask, nz, sigma, thr, values, model, counts, lengths, lhist, total)


# ------------------------------------------------------------------ several layers at once
# A fit is a chain of short dependent launches (a Lloyd iteration on a value-sorted vector touches a few kilobytes), so one
# layer keeps a handful of the 256 CUs busy; the layers of a model are independent (Trainer.quantize walks them one after the
# other, common/trainer.py:42-72), so several of them run side by side: one host thread and one HIP stream per worker.
_POOLS: dict = {}
_POOL_LOCK = threading.Lock()
_WORKER = threading.local()


def _pool(workers: int) -> ThreadPoolExecutor:
    with _POOL_LOCK:
        p = _POOLS.get(workers)
        if p is None:
            # kept for the life of the process: the workers' pinned landing zones and streams are made once
            p = _POOLS[workers] = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="nnc-layer")
        return p


def _worker_stream(dev) -> torch.cuda.Stream:
    st = getattr(_WORKER, "streams", None)
    if st is None:
        st = _WORKER.streams = {}
    if dev not in st:
        st[dev] = torch.cuda.Stream(device=dev)
    return st[dev]

def _compress_layers_local(layers, workers: int = 8, **kw):
    """compress_layer for every tensor of ``layers`` (device float32 tensors, each pruned in place when ``q`` is given),
    ``workers`` of them side by side on streams of their own; the results come back in the order of ``layers`` and are the ones
    the calls would give one after the other.  Modes that draw from NumPy's global generator (forgy) run one after the other,
    so that the draws stay in layer order like the reference's.  ``kw``: the arguments of compress_layer (single GPU)."""
    layers = list(layers)
    if workers <= 1 or len(layers) <= 1 or kw.get("mode", "linear") == "forgy":
        return [compress_layer(t, **kw) for t in layers]
    dev = layers[0].device
    main = torch.cuda.current_stream(dev)
    ready = torch.cuda.Event()
    ready.record(main)
    order = sorted(range(len(layers)), key=lambda i: -layers[i].numel())   # long ones first: the tail of the schedule is short ones
    nxt = iter(order)
    lock = threading.Lock()
    results = [None] * len(layers)

    def work():
        torch.cuda.set_device(dev)
        s = _worker_stream(dev)
        s.wait_event(ready)
        with torch.cuda.stream(s):
            while True:
                with lock:
                    i = next(nxt, None)
                if i is None:
                    break
                results[i] = compress_layer(layers[i], **kw)
        done = torch.cuda.Event()
        done.record(s)
        return done

    futures = [_pool(workers).submit(work) for _ in range(min(workers, len(layers)))]
    for f in futures:
        main.wait_event(f.result())
    for r in results:   # made on a worker's stream, used from here on on the caller's
        for t in (r.mask, r.values, None if r.model is None else r.model.labels_compact_, None if r.model is None else r.model.counts_device_):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)
    return results


# ------------------------------------------------------------------ the layers of a model over the GPUs of a node
@dataclass
class LayerRecord:
    """What every rank knows about every layer after compress_layers(group=...): the K-sized results (a few hundred bytes a
    layer; the index and value vectors stay on the rank that made them, in ``result``)."""
    index: int
    n: int
    rank: int                       # owner, or -1: sharded over all ranks of the group
    nzeroed: int | None
    sigma: float | None
    threshold: float | None
    centers: np.ndarray | None      # float32[K], None if the tensor passed through ("not enough bits")
    counts: np.ndarray | None
    code_lengths: np.ndarray | None
    total_bits: int | None
    n_iter: int
    n_relocations: int
    stop: str | None
    result: LayerResult | None = None   # this rank's LayerResult (owner, or this rank's shard); None on the other ranks


def layer_cost(n: int) -> float:
    """Seconds one GPU spends on a tensor of n weights, to the accuracy a schedule needs: a fit is a chain of short dependent
    launches whatever the length (some 0.35 ms for the 4-bit fits of BASELINE configs[4]), the passes over the vector add
    about 0.1 ns a weight."""
    return 0.35e-3 + 1.0e-10 * int(n)


def partition_layers(sizes, world: int, shard_above: int | None = None, cost=layer_cost):
    """owner[i] for every tensor: longest processing time first onto the least loaded rank (ties: the lowest rank, the lowest
    index first -- every rank computes the same table from the sizes alone); -1 for tensors of ``shard_above`` weights or
    more, which all ranks work on together as shards."""
    sizes = [int(s) for s in sizes]
    owner = [0] * len(sizes)
    load = [0.0] * world
    for i in sorted(range(len(sizes)), key=lambda i: (-cost(sizes[i]), i)):
        if shard_above is not None and sizes[i] >= shard_above and world > 1:
            owner[i] = -1
            continue
        r = min(range(world), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += cost(sizes[i])
    return owner


def _record(i, n, rank, r: LayerResult) -> LayerRecord:
    m = r.model
    return LayerRecord(i, int(n), rank, r.nzeroed, r.sigma, r.threshold,
                       None if m is None else np.asarray(m.cluster_centers_, dtype=np.float32).ravel().copy(),
                       r.counts, r.code_lengths, r.total_bits, 0 if m is None else int(m.n_iter_),
                       0 if m is None else int(m.n_relocations_), None if m is None else m.stop_reason_, r)


def _compress_layers_ranks(layers, sizes, workers, group, shard_above, comm, _compress, kw):
    import copy

    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if kw.get("mode", "linear") in ("forgy", "kmeans++"):
        raise ValueError("compress_layers(group=...): modes that draw from NumPy's global generator (forgy, kmeans++) consume it in layer "
                         "order (common/trainer.py:50-70) and cannot be dealt out to ranks; run them on one rank")
    if sizes is None:
        if any(t is None for t in layers):
            raise ValueError("compress_layers(group=...): `sizes` (the length of EVERY tensor, the same list on every rank) is needed "
                             "when this rank holds only its own tensors")
        sizes = [t.numel() for t in layers]
    if len(sizes) != len(layers):
        raise ValueError(f"compress_layers: {len(layers)} layers but {len(sizes)} sizes")
    owner = partition_layers(sizes, world, shard_above)
    records = {}
    # tensors every rank holds a shard of: all ranks together, one after the other in index order (each is a chain of collectives)
    for i in [i for i, o in enumerate(owner) if o < 0]:
        if layers[i] is None:
            raise ValueError(f"compress_layers: tensor {i} ({sizes[i]} weights) is sharded over the group; this rank's shard is missing")
        r = _compress(layers[i], group=group, comm=comm, **kw)
        records[i] = _record(i, sizes[i], -1, r)
    mine = [i for i, o in enumerate(owner) if o == rank]
    for i in mine:
        if layers[i] is None or layers[i].numel() != sizes[i]:
            raise ValueError(f"compress_layers: tensor {i} belongs to rank {rank} (partition_layers) and must be given whole "
                             f"({sizes[i]} weights)")
    if _compress is compress_layer:
        res = _compress_layers_local([layers[i] for i in mine], workers=workers, **kw)
    else:
        res = [_compress(layers[i], **kw) for i in mine]
    for i, r in zip(mine, res):
        records[i] = _record(i, sizes[i], rank, r)
    # every rank learns every layer's K-sized results: one object gather (a few hundred bytes a layer, no data-path collective)
    wire = []
    for i in mine:
        c = copy.copy(records[i])
        c.result = None
        wire.append(c)
    gathered = [None] * world
    dist.all_gather_object(gathered, wire, group=group)
    for part in gathered:
        for rec in part:
            if rec.index not in records:
                records[rec.index] = rec
    missing = [i for i in range(len(sizes)) if i not in records]
    if missing:
        raise RuntimeError(f"compress_layers: no rank reported tensors {missing[:8]}")
    return [records[i] for i in range(len(sizes))]


def compress_layers(layers, workers: int = 8, group=None, sizes=None, shard_above: int | None = None, comm=None, _compress=None, **kw):
    """compress_layer for every tensor of a model (Trainer.quantize's loop, common/trainer.py:50-70).

    One GPU (``group=None``): ``workers`` tensors side by side on streams of their own (see _compress_layers_local); returns the
    LayerResults in the order of ``layers``.

    One process per GPU (``group``: torch.distributed group): the tensors are dealt out to the ranks by partition_layers (longest
    first onto the least loaded rank; replicas, no collective on the data path), each rank runs its own ones as above, and
    every rank gets the list of LayerRecords of ALL tensors (K-sized results; ``.result`` holds the device tensors on the rank
    that made them).  ``layers[i]`` may be None for tensors of other ranks (then pass ``sizes``, the lengths of all tensors).
    Tensors of ``shard_above`` weights or more are not dealt out but sharded: every rank passes its shard
    (sharding.shard_bounds) and they go through the sharded fit (``comm``: sharding.RcclComm) -- same results either way.
    ``_compress``: the per-tensor function (tests put the CPU oracle here to rehearse the dealing without a GPU)."""
    layers = list(layers)
    if group is None:
        if _compress is not None and _compress is not compress_layer:
            return [_compress(t, **kw) for t in layers]
        return _compress_layers_local(layers, workers=workers, **kw)
    return _compress_layers_ranks(layers, sizes, workers, group, shard_above, comm, _compress or compress_layer, kw)

