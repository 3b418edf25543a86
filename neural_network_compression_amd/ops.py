"""Device-level operators: thin torch-tensor wrappers over the C ABI (include/nnc.h).

Every function takes CUDA (ROCm) tensors that are already resident in HBM, enqueues HIP
kernels from csrc/nnc_hip.hip on torch's current stream and returns device tensors; none
of them synchronises unless its docstring says so.  PyTorch is plumbing here (memory,
streams); the arithmetic is in the HIP kernels.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import threading
import math

import numpy as np
import torch

from . import _native as nat


def _require_cuda(t: torch.Tensor, name: str, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA (ROCm) torch tensor resident in HBM")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


_STAGE = threading.local()
_STAGE_SLOTS, _STAGE_BYTES = 16, 8192


def small_to_device(arr: np.ndarray, dev) -> torch.Tensor:
    """A few kilobytes of host data (initial centres, histogram steps) to the device without the blocking staging copy a pageable
    source costs: through a ring of pinned slots of this host thread, asynchronously on the current stream.  A slot is reused
    only after the copy that last read it has completed (an event per slot)."""
    arr = np.ascontiguousarray(arr)
    nb = arr.nbytes
    if nb == 0 or nb > _STAGE_BYTES:
        return torch.from_numpy(arr).to(dev)
    ring = getattr(_STAGE, "ring", None)
    if ring is None:
        ring = _STAGE.ring = {"buf": torch.empty(_STAGE_SLOTS * _STAGE_BYTES, dtype=torch.uint8, pin_memory=True),
                              "ev": [None] * _STAGE_SLOTS, "next": 0}
    i = ring["next"]
    ring["next"] = (i + 1) % _STAGE_SLOTS
    if ring["ev"][i] is not None:
        ring["ev"][i].synchronize()
    slot = ring["buf"][i * _STAGE_BYTES: i * _STAGE_BYTES + nb]
    slot.numpy()[:] = arr.view(np.uint8).reshape(-1)
    out = torch.empty(nb, dtype=torch.uint8, device=dev)
    out.copy_(slot, non_blocking=True)
    ev = ring["ev"][i] = ring["ev"][i] or torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    return out.view(torch.from_numpy(arr).dtype).reshape(arr.shape)


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def device_info():
    L = nat.load()
    buf = ctypes.create_string_buffer(64)
    cu = ctypes.c_int(0)
    nat.check(L.nnc_device_info(buf, 64, ctypes.byref(cu)))
    return buf.value.decode(), cu.value


# ------------------------------------------------------------------ NumPy-exact reductions
def chunk_sums(x: torch.Tensor, mean_dev: torch.Tensor | None = None) -> torch.Tensor:
    """Per-8192-chunk NumPy pairwise sums of x (or of (x-mean)^2 when mean_dev is given)."""
    _require_cuda(x, "x", torch.float32)
    L = nat.load()
    n = x.numel()
    out = torch.empty((n + nat.NNC_CHUNK - 1) // nat.NNC_CHUNK, dtype=torch.float32, device=x.device)
    nat.check(L.nnc_chunk_sums_f32(_ptr(x), n, 1 if mean_dev is not None else 0, _ptr(mean_dev), _ptr(out), _stream(x)))
    return out


def fold(chunks: torch.Tensor, count: int, op: int, scale_dev: torch.Tensor | None = None) -> torch.Tensor:
    """Left-to-right float32 fold; returns a device float32[2] = {result, result*scale}."""
    _require_cuda(chunks, "chunks", torch.float32)
    L = nat.load()
    out = torch.empty(2, dtype=torch.float32, device=chunks.device)   # [1] is written only with a scale; nobody reads it otherwise
    nat.check(L.nnc_fold_f32(_ptr(chunks), chunks.numel(), int(count), op, _ptr(scale_dev), _ptr(out), _stream(chunks)))
    return out


def _gather_chunks(chunks: torch.Tensor, group) -> torch.Tensor:
    from . import sharding

    return sharding.gather_chunks(chunks, group)


def moments(x: torch.Tensor, n_total: int | None = None, group=None):
    """NumPy-exact float32 (mean, var, std) of the whole vector, as device tensors.

    Sharded (``group`` given): every rank passes its shard, shards start on multiples of
    8192 elements; the chunk sums are all-gathered and folded identically on every rank."""
    n_total = x.numel() if n_total is None else int(n_total)
    c1 = chunk_sums(x)
    if group is not None:
        c1 = _gather_chunks(c1, group)
    mean = fold(c1, n_total, nat.FOLD_MEAN)
    c2 = chunk_sums(x, mean)
    if group is not None:
        c2 = _gather_chunks(c2, group)
    var = fold(c2, n_total, nat.FOLD_MEAN)
    std = fold(c2, n_total, nat.FOLD_STD)
    return mean[0:1], var[0:1], std[0:1]


# ------------------------------------------------------------------ prune
def prune_(x: torch.Tensor, q: float, std_smooth: bool = True):
    """In place: mask = |x| < (std(x)*q | q); x[mask] = 0.  Returns (mask uint8, stats float32[2]
    = {sigma, thr}, nzeroed int64[1]) as device tensors.  q is taken as float32."""
    _require_cuda(x, "x", torch.float32)
    L = nat.load()
    n = x.numel()
    mask = torch.empty(n, dtype=torch.uint8, device=x.device)
    # (the library writes both: stats = {sigma, threshold} with std_smooth, {-, threshold} without; the count is zeroed there)
    stats = torch.empty(2, dtype=torch.float32, device=x.device) if std_smooth else torch.zeros(2, dtype=torch.float32, device=x.device)
    nz = torch.empty(1, dtype=torch.int64, device=x.device)
    ws_bytes = L.nnc_prune_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    nat.check(L.nnc_prune_f32(_ptr(x), n, float(np.float32(q)), 1 if std_smooth else 0, _ptr(mask), _ptr(stats),
                              _ptr(nz), _ptr(ws), ws_bytes, _stream(x)))
    return mask.view(x.shape), stats, nz


def prune_stats_(x: torch.Tensor, q: float, std_smooth: bool = True):
    """prune_ and, from the same pass, minmax_signs of the pruned tensor: (mask, stats, nzeroed, float32[4] = {min, max, min
    over the non-zeros, max over the non-zeros}, int64[2] = {#negative, #zero})."""
    _require_cuda(x, "x", torch.float32)
    if x.numel() == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    L = nat.load()
    n = x.numel()
    mask = torch.empty(n, dtype=torch.uint8, device=x.device)
    stats = torch.empty(2, dtype=torch.float32, device=x.device) if std_smooth else torch.zeros(2, dtype=torch.float32, device=x.device)
    nz = torch.empty(1, dtype=torch.int64, device=x.device)
    mm = torch.empty(4, dtype=torch.float32, device=x.device)
    signs = torch.empty(2, dtype=torch.int64, device=x.device)
    ws_bytes = L.nnc_prune_stats_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    nat.check(L.nnc_prune_stats_f32(_ptr(x), n, float(np.float32(q)), 1 if std_smooth else 0, _ptr(mask), _ptr(stats), _ptr(nz), _ptr(mm), _ptr(signs),
                                    _ptr(ws), ws_bytes, _stream(x)))
    return mask.view(x.shape), stats, nz, mm, signs


def threshold_mask_(x: torch.Tensor, thr_dev: torch.Tensor):
    """In place threshold pass with the float32 threshold already on the device."""
    _require_cuda(x, "x", torch.float32)
    _require_cuda(thr_dev, "thr_dev", torch.float32)
    L = nat.load()
    n = x.numel()
    mask = torch.empty(n, dtype=torch.uint8, device=x.device)
    nz = torch.empty(1, dtype=torch.int64, device=x.device)   # zeroed by the library
    nat.check(L.nnc_threshold_mask_f32(_ptr(x), n, _ptr(thr_dev), _ptr(mask), _ptr(nz), _stream(x)))
    return mask.view(x.shape), nz


def apply_mask_(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """x[mask] = 0 in place (Trainer._reset_pruned_parameters)."""
    _require_cuda(x, "x", torch.float32)
    _require_cuda(mask, "mask")
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    if mask.dtype != torch.uint8 or mask.numel() != x.numel():
        raise ValueError("mask must be uint8/bool with as many elements as x")
    L = nat.load()
    nat.check(L.nnc_apply_mask_f32(_ptr(x), _ptr(mask), x.numel(), _stream(x)))
    return x


# ------------------------------------------------------------------ distribution passes
def minmax(x: torch.Tensor, skip_zeros: bool = False):
    """(float32[2] = {min, max}, int64[1] count) on the device."""
    _require_cuda(x, "x", torch.float32)
    if x.numel() == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    L = nat.load()
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    cnt = torch.empty(1, dtype=torch.int64, device=x.device)   # written by the final reduction
    ws_bytes = L.nnc_minmax_workspace_bytes(x.numel())
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    nat.check(L.nnc_minmax_f32(_ptr(x), x.numel(), 1 if skip_zeros else 0, _ptr(out), _ptr(cnt), _ptr(ws), ws_bytes, _stream(x)))
    return out, cnt


def minmax_signs(x: torch.Tensor):
    """(float32[4] = {min, max, min over the non-zeros, max over the non-zeros}, int64[2] = {#negative, #zero})
    on the device, one pass."""
    _require_cuda(x, "x", torch.float32)
    if x.numel() == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    L = nat.load()
    out = torch.empty(4, dtype=torch.float32, device=x.device)
    signs = torch.empty(2, dtype=torch.int64, device=x.device)
    ws_bytes = L.nnc_minmax_workspace_bytes(x.numel())
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    nat.check(L.nnc_minmax_signs_f32(_ptr(x), x.numel(), _ptr(out), _ptr(signs), _ptr(ws), ws_bytes, _stream(x)))
    return out, signs


def hist31(x: torch.Tensor, steps: torch.Tensor, skip_zeros: bool = False) -> torch.Tensor:
    """int64[31] counts of steps[b] <= x < steps[b+1] (steps: device float32[32])."""
    _require_cuda(x, "x", torch.float32)
    _require_cuda(steps, "steps", torch.float32)
    if steps.numel() != 32:
        raise ValueError("steps must hold 32 values")
    L = nat.load()
    counts = torch.zeros(31, dtype=torch.int64, device=x.device)
    nat.check(L.nnc_hist31_f32(_ptr(x), x.numel(), 1 if skip_zeros else 0, _ptr(steps), _ptr(counts), _stream(x)))
    return counts


def bincount(labels: torch.Tensor, k: int) -> torch.Tensor:
    """int64[k] histogram of uint8 / uint16(int16 storage) centroid indices."""
    _require_cuda(labels, "labels")
    if labels.dtype == torch.uint8:
        lb = 1
    elif labels.dtype in (torch.int16, torch.uint16):
        lb = 2
    else:
        raise TypeError("labels must be uint8 or 16-bit")
    L = nat.load()
    counts = torch.zeros(int(k), dtype=torch.int64, device=labels.device)
    nat.check(L.nnc_bincount(_ptr(labels), lb, labels.numel(), int(k), _ptr(counts), _stream(labels)))
    return counts


def _label_bytes(labels: torch.Tensor) -> int:
    if labels.dtype == torch.uint8:
        return 1
    if labels.dtype in (torch.int16, torch.uint16):
        return 2
    raise TypeError("labels must be uint8 or 16-bit (QuantizedModel.labels_compact_)")


def centroid_gradient(grad: torch.Tensor, labels: torch.Tensor, k: int, group=None) -> torch.Tensor:
    """dL/dC_j = sum of dL/dW over the weights whose centroid index is j (Deep Compression's centroid fine-tuning,
    described but left out by the reference, papers/lat/report.tex:149-158) -> float64[k] on the device.
    Exact fixed-point sums (include/nnc.h, nnc_centroid_grad_f32): independent of order and, with ``group`` (every rank
    holds a shard of the layer), of the number of GPUs.  One host read (max |grad|) sizes the fixed point."""
    _require_cuda(grad, "grad", torch.float32)
    _require_cuda(labels, "labels")
    g = grad.reshape(-1)
    if labels.numel() != g.numel():
        raise ValueError("labels and grad must have the same number of elements")
    L = nat.load()
    n = g.numel()
    if n == 0:
        return torch.zeros(int(k), dtype=torch.float64, device=g.device)
    mm, _ = minmax(g)
    n_total = n
    if group is not None:
        from . import sharding

        mm = sharding.allreduce_minmax(mm, group)
        n_total = sharding.total_count(n, g.device, group)
    host = mm.cpu().numpy()
    S = fix_shift(float(max(abs(host[0]), abs(host[1]))), n_total)
    sums = torch.empty(int(k), dtype=torch.int64, device=g.device)
    nat.check(L.nnc_centroid_grad_f32(_ptr(g), _ptr(labels), _label_bytes(labels), n, int(k), S, _ptr(sums), 0, _stream(g)))
    if group is not None:
        from . import sharding

        sharding.allreduce_sum_(sums, group)
    return torch.ldexp(sums.to(torch.float64), torch.tensor(-S, device=g.device))


def gather(centers: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """cluster_centers_[labels_] on the device (utility.py:239): float32 vector of labels.numel() values."""
    _require_cuda(centers, "centers", torch.float32)
    _require_cuda(labels, "labels")
    L = nat.load()
    out = torch.empty(labels.numel(), dtype=torch.float32, device=labels.device)
    nat.check(L.nnc_gather_f32(_ptr(centers), centers.numel(), _ptr(labels), _label_bytes(labels), labels.numel(), _ptr(out), _stream(labels)))
    return out


def huffman_lengths(counts) -> tuple:
    """Host: (lengths uint8[k], hist int64[max_len+1], total_bits) from an index histogram."""
    L = nat.load()
    c = np.ascontiguousarray(np.asarray(counts, dtype=np.int64))
    k = c.size
    lengths = np.zeros(k, dtype=np.uint8)
    hist = np.zeros(65, dtype=np.int64)
    total = ctypes.c_int64(0)
    nat.check(L.nnc_huffman_lengths(c.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), k,
                                    lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                    hist.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(total)))
    top = int(lengths.max()) if k else 0
    return lengths, hist[: top + 1].copy(), int(total.value)


# ------------------------------------------------------------------ fixed-point rule (host mirror)
def fix_shift(absmax: float, n_total: int) -> int:
    """S of the fixed-point sums (same rule as nnc_fix_shift; pure host arithmetic)."""
    L = max(1, (int(n_total) - 1).bit_length())
    if not (absmax > 0) or not math.isfinite(absmax):
        return 0
    _, P = math.frexp(float(absmax))
    return min(28, 62 - L) - P


def fix_f32(v, S: int) -> int:
    """Fixed-point image of one float32, (int) rint(v * 2^S) with ties to even (bit-for-bit the
    device function fix_f32)."""
    from fractions import Fraction

    fr = Fraction(float(np.float32(v))) * (Fraction(2) ** int(S))
    fl = fr.numerator // fr.denominator
    rem = fr - fl
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
        fl += 1
    return int(fl)
