"""Compressed on-disk form of quantized layers: codebook + Huffman-coded centroid indices, dense or relative-index sparse.

The reference keeps nothing on disk and only names the third stage of Deep Compression (Huffman coding,
README.md:9); what ``get_quantized_weight`` returns -- ``cluster_centers_`` and ``labels_``
(neural_network_compression/common/utility.py:239) -- is exactly a codebook and an index stream, so the stored layer is
those two, with the indices entropy coded on the GPU (include/nnc.h, nnc_huffman_*; csrc/nnc_codec.hip).  Decoding gives
back ``cluster_centers_[labels_]`` bit for bit.

Two forms of the index stream, the smaller one is kept per tensor (``form="auto"``):
  * dense  : every index Huffman coded (the pruned zeros share one centroid, whose index then costs one bit);
  * sparse : Deep Compression's relative-index format (section 3 of the paper the reference's report follows,
             papers/lat/report.tex:327): only the indices that are not the zero cluster's, each with the distance to the
             previous stored position in 4 or 8 bits, filler entries for longer gaps; both entry streams Huffman coded.

File layout (little endian), one record per tensor:

    "NNC2" | u32 n_tensors
    per tensor: u16 name length, name (utf-8) | u8 ndim, u64 shape[ndim] | u32 K | u8 label_bytes | u64 N | u64 total_bits
                f32 codebook[K] | u8 form (0 dense, 1 sparse)
      dense :   STREAM(K, N)
      sparse:   u8 delta_bits | u32 zero_symbol | u64 entries | u16 entries_in_chunk[ceil(N / 1024)]
                STREAM(2^delta_bits, entries) of the distances - 1 | STREAM(K, entries) of the indices
      STREAM(k, n) = u64 bits | u8 code_length[k] | u32 chunk_bits[ceil(n / 1024)] | u32 words[ceil(bits / 32)]  (MSB-first)
    total_bits = all stream bits of the record (what the compression ratio counts besides the tables).
    a tensor that passed through unquantized ("not enough bits") is stored raw: K = 0, then f32 data[N].
"""
from __future__ import annotations

import ctypes
import struct
from typing import Dict, Tuple

import numpy as np
import torch

from . import _native as nat
from . import ops

MAGIC = b"NNC2"
CHUNK = 1024


def _flatten_lengths(lengths: np.ndarray, counts: np.ndarray) -> np.ndarray:
    """Code lengths beyond 32 bits (possible only for hugely skewed histograms with hundreds of symbols) are replaced by a
    fixed-width code over the used symbols: still a prefix code, decodable by the same tables."""
    if int(lengths.max(initial=0)) <= 32:
        return lengths
    used = counts > 0
    width = max(1, int(np.ceil(np.log2(max(2, int(used.sum()))))))
    out = np.zeros_like(lengths)
    out[used] = width
    return out


def encode_indices(labels: torch.Tensor, k: int, counts: np.ndarray | None = None):
    """labels: device uint8 / int16-storage centroid indices.  Returns (words uint32 device tensor, chunk_bits np.uint32[nchunks],
    lengths np.uint8[k], total_bits)."""
    L = nat.load()
    n = labels.numel()
    lb = ops._label_bytes(labels)
    if counts is None:
        counts = ops.bincount(labels, k).cpu().numpy()
    counts = np.asarray(counts, dtype=np.int64)
    lengths, _, _ = ops.huffman_lengths(counts)
    lengths = _flatten_lengths(np.ascontiguousarray(lengths, dtype=np.uint8), counts)
    codes = np.zeros(k, dtype=np.uint32)
    nat.check(L.nnc_huffman_codes(lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), k, codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
    dev = labels.device
    stream = ops._stream(labels)
    len_d = torch.from_numpy(lengths).to(dev)
    codes_d = torch.from_numpy(codes.view(np.int32)).to(dev)
    nchunks = int(L.nnc_codec_chunks(n))
    off = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
    nat.check(L.nnc_huffman_chunk_offsets(ops._ptr(labels), lb, n, len_d.data_ptr(), k, off.data_ptr(), stream))
    total_bits = int((counts * lengths.astype(np.int64)).sum())     # known on the host: no read of the device total needed
    nwords = total_bits // 32 + 2
    words = torch.empty(nwords, dtype=torch.int32, device=dev)
    nat.check(L.nnc_huffman_encode(ops._ptr(labels), lb, n, codes_d.data_ptr(), len_d.data_ptr(), k, off.data_ptr(), words.data_ptr(), nwords, stream))
    off_h = off.cpu().numpy()
    assert int(off_h[-1]) == total_bits, (int(off_h[-1]), total_bits)
    chunk_bits = np.diff(off_h).astype(np.uint32)
    return words[: (total_bits + 31) // 32], chunk_bits, lengths, total_bits


def decode_indices(words: torch.Tensor, chunk_bits: np.ndarray, n: int, lengths: np.ndarray, k: int, label_bytes: int) -> torch.Tensor:
    """The inverse of encode_indices on the device; raises if the stream does not parse."""
    L = nat.load()
    dev = words.device
    stream = ops._stream(words)
    tb = int(L.nnc_huffman_decode_tables_bytes())
    tables = np.zeros(tb, dtype=np.uint8)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint8)
    nat.check(L.nnc_huffman_decode_tables(lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), k, tables.ctypes.data, tb))
    tables_d = torch.from_numpy(tables).to(dev)
    off = np.zeros(chunk_bits.size + 1, dtype=np.int64)
    np.cumsum(chunk_bits.astype(np.int64), out=off[1:])
    off_d = torch.from_numpy(off).to(dev)
    padded = torch.zeros(words.numel() + 2, dtype=torch.int32, device=dev)   # the decoder may look one word past the end
    padded[: words.numel()] = words
    out = torch.empty(n, dtype=torch.uint8 if label_bytes == 1 else torch.int16, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    nat.check(L.nnc_huffman_decode(padded.data_ptr(), off_d.data_ptr(), n, tables_d.data_ptr(), k, ops._ptr(out), label_bytes, bad.data_ptr(), stream))
    if int(bad.item()):
        raise ValueError("corrupt index stream")
    return out


def encode_sparse(labels: torch.Tensor, zero_symbol: int, delta_bits: int):
    """The relative-index entries of ``labels`` (device) -> (delta uint8 device tensor [distance - 1], sym device tensor [indices, the
    labels' width], entries_in_chunk np.uint16[nchunks])."""
    L = nat.load()
    n = labels.numel()
    lb = ops._label_bytes(labels)
    dev = labels.device
    stream = ops._stream(labels)
    nchunks = int(L.nnc_codec_chunks(n))
    off = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
    nat.check(L.nnc_sparse_entry_offsets(ops._ptr(labels), lb, n, int(zero_symbol), int(delta_bits), off.data_ptr(), stream))
    off_h = off.cpu().numpy()
    entries = int(off_h[-1])
    delta = torch.empty(max(entries, 1), dtype=torch.uint8, device=dev)
    sym = torch.empty(max(entries, 1), dtype=labels.dtype, device=dev)
    nat.check(L.nnc_sparse_emit(ops._ptr(labels), lb, n, int(zero_symbol), int(delta_bits), off.data_ptr(), delta.data_ptr(), sym.data_ptr(), stream))
    return delta[:entries], sym[:entries], np.diff(off_h).astype(np.uint16)


def decode_sparse(delta: torch.Tensor, sym: torch.Tensor, entries_in_chunk: np.ndarray, n: int, zero_symbol: int) -> torch.Tensor:
    """The inverse of encode_sparse on the device; raises if an entry points outside its chunk."""
    L = nat.load()
    dev = sym.device
    off = np.zeros(entries_in_chunk.size + 1, dtype=np.int64)
    np.cumsum(entries_in_chunk.astype(np.int64), out=off[1:])
    if int(off[-1]) != delta.numel() or delta.numel() != sym.numel():
        raise ValueError("corrupt entry table")
    off_d = torch.from_numpy(off).to(dev)
    out = torch.empty(n, dtype=sym.dtype, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    nat.check(L.nnc_sparse_expand(ops._ptr(delta), ops._ptr(sym), ops._label_bytes(sym), off_d.data_ptr(), n, int(zero_symbol), out.data_ptr(), bad.data_ptr(),
                                  ops._stream(sym)))
    if int(bad.item()):
        raise ValueError("corrupt sparse entries")
    return out


def _stream_bytes(words, chunk_bits, lengths, total_bits) -> bytes:
    return struct.pack("<Q", int(total_bits)) + lengths.tobytes() + chunk_bits.tobytes() + words.cpu().numpy().tobytes()


def _read_stream(blob, pos, k, n, device, label_bytes):
    (total_bits,) = struct.unpack_from("<Q", blob, pos); pos += 8
    lengths = np.frombuffer(blob, dtype=np.uint8, count=k, offset=pos); pos += k
    nchunks = (n + CHUNK - 1) // CHUNK
    chunk_bits = np.frombuffer(blob, dtype=np.uint32, count=nchunks, offset=pos); pos += 4 * nchunks
    nwords = (total_bits + 31) // 32
    words = np.frombuffer(blob, dtype=np.int32, count=nwords, offset=pos); pos += 4 * nwords
    if int(chunk_bits.astype(np.int64).sum()) != total_bits:
        raise ValueError("corrupt chunk table")
    if n == 0:
        return torch.empty(0, dtype=torch.uint8 if label_bytes == 1 else torch.int16, device=device), pos
    return decode_indices(torch.from_numpy(words.copy()).to(device), chunk_bits, n, lengths.copy(), k, label_bytes), pos


SPARSE_MIN_ZERO_SHARE = 0.5     # below this share of zero-cluster indices the sparse form cannot win: it is not even tried


def pack_indices(labels: torch.Tensor, k: int, counts: np.ndarray | None = None, form: str = "auto"):
    """The index stream of one tensor -> (bytes from the `form` byte on, total stream bits, form name).  form: "dense", "sparse4",
    "sparse8", or "auto" (the smallest of the three in bytes)."""
    n = labels.numel()
    if counts is None:
        counts = ops.bincount(labels, k).cpu().numpy()
    counts = np.asarray(counts, dtype=np.int64)
    cands = {}
    if form in ("dense", "auto"):
        words, chunk_bits, lengths, total_bits = encode_indices(labels, k, counts)
        cands["dense"] = (struct.pack("<B", 0) + _stream_bytes(words, chunk_bits, lengths, total_bits), total_bits)
    zero = int(np.argmax(counts))
    for name, db in (("sparse4", 4), ("sparse8", 8)):
        if form != name and not (form == "auto" and n > 0 and counts[zero] >= SPARSE_MIN_ZERO_SHARE * n):
            continue
        delta, sym, per_chunk = encode_sparse(labels, zero, db)
        e = delta.numel()
        body = struct.pack("<BBIQ", 1, db, zero, e) + per_chunk.tobytes()
        bits = 0
        for arr, kk in ((delta, 1 << db), (sym, k)):
            if e:
                w, cb, ln, tb = encode_indices(arr, kk)
            else:
                w, cb, ln, tb = torch.empty(0, dtype=torch.int32, device=labels.device), np.zeros(0, np.uint32), np.zeros(kk, np.uint8), 0
            body += _stream_bytes(w, cb, ln, tb)
            bits += tb
        cands[name] = (body, bits)
    if not cands:
        raise ValueError(f"unknown index form {form!r}")
    best = min(cands, key=lambda nm: (len(cands[nm][0]), nm))
    return cands[best][0], cands[best][1], best


def unpack_indices(blob, pos, k, n, lb, device):
    (form,) = struct.unpack_from("<B", blob, pos); pos += 1
    if form == 0:
        return _read_stream(blob, pos, k, n, device, lb)
    if form != 1:
        raise ValueError("unknown index form")
    db, zero, e = struct.unpack_from("<BIQ", blob, pos); pos += 13
    nchunks = (n + CHUNK - 1) // CHUNK
    per_chunk = np.frombuffer(blob, dtype=np.uint16, count=nchunks, offset=pos); pos += 2 * nchunks
    delta, pos = _read_stream(blob, pos, 1 << db, e, device, 1)
    sym, pos = _read_stream(blob, pos, k, e, device, lb)
    return decode_sparse(delta, sym, per_chunk, n, zero), pos


def pack_tensor(name: str, shape, model, raw: torch.Tensor | None = None, form: str = "auto", info: dict | None = None) -> bytes:
    """One record.  model: kmeans.QuantizedModel (or None with ``raw`` = the unquantized float32 tensor).  ``info`` (optional dict)
    receives what a report needs: n, bytes, stream bits, form."""
    nm = name.encode("utf-8")
    head = struct.pack("<H", len(nm)) + nm + struct.pack("<B", len(shape)) + b"".join(struct.pack("<Q", int(d)) for d in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if model is None:
        data = np.ascontiguousarray(raw.detach().cpu().numpy(), dtype=np.float32).reshape(-1)
        rec = head + struct.pack("<IBQQ", 0, 0, n, 0) + data.tobytes()
        if info is not None:
            info.update({"n": n, "bytes": len(rec), "stream_bits": 32 * n, "form": "raw", "k": 0})
        return rec
    k = int(model.cluster_centers_.size)
    labels = model.labels_compact_
    counts = model.counts_device_.cpu().numpy() if getattr(model, "counts_device_", None) is not None else None
    body_idx, total_bits, chosen = pack_indices(labels, k, counts, form)
    lb = ops._label_bytes(labels)
    body = struct.pack("<IBQQ", k, lb, n, total_bits)
    body += np.ascontiguousarray(model.cluster_centers_.ravel(), dtype=np.float32).tobytes()
    rec = head + body + body_idx
    if info is not None:
        info.update({"n": n, "bytes": len(rec), "stream_bits": int(total_bits), "form": chosen, "k": k})
    return rec


def save_compressed(path: str, tensors: Dict[str, Tuple[tuple, object, torch.Tensor | None]], form: str = "auto", report: dict | None = None) -> int:
    """tensors: name -> (shape, QuantizedModel | None, raw tensor if unquantized).  Returns the file size in bytes; ``report`` (optional
    dict) receives name -> {n, bytes, stream_bits, form, k} and, under "total", the whole file against 32 bits per weight."""
    blob = MAGIC + struct.pack("<I", len(tensors))
    total_n = 0
    for name, (shape, model, raw) in tensors.items():
        info = {}
        blob += pack_tensor(name, tuple(shape), model, raw, form, info)
        total_n += info["n"]
        if report is not None:
            info["bits_per_weight"] = 8.0 * info["bytes"] / max(info["n"], 1)
            report[name] = info
    with open(path, "wb") as f:
        f.write(blob)
    if report is not None:
        report["total"] = {"n": total_n, "bytes": len(blob), "bits_per_weight": 8.0 * len(blob) / max(total_n, 1),
                           "compression_ratio": 4.0 * total_n / max(len(blob), 1)}
    return len(blob)


def load_compressed(path: str, device=None) -> Dict[str, torch.Tensor]:
    """name -> decoded float32 tensor on the device: cluster_centers_[labels_] in the stored shape."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:4] != MAGIC:
        raise ValueError("not an NNC2 file")
    (nt,) = struct.unpack_from("<I", blob, 4)
    pos = 8
    out = {}
    for _ in range(nt):
        (ln,) = struct.unpack_from("<H", blob, pos); pos += 2
        name = blob[pos: pos + ln].decode("utf-8"); pos += ln
        (nd,) = struct.unpack_from("<B", blob, pos); pos += 1
        shape = struct.unpack_from("<" + "Q" * nd, blob, pos); pos += 8 * nd
        k, lb, n, total_bits = struct.unpack_from("<IBQQ", blob, pos); pos += 21
        if k == 0:
            data = np.frombuffer(blob, dtype=np.float32, count=n, offset=pos); pos += 4 * n
            out[name] = torch.from_numpy(data.copy()).to(device).reshape(shape)
            continue
        centers = np.frombuffer(blob, dtype=np.float32, count=k, offset=pos); pos += 4 * k
        labels, pos = unpack_indices(blob, pos, k, n, lb, device)
        out[name] = ops.gather(torch.from_numpy(centers.copy()).to(device), labels).reshape(shape)
    return out
